// Developer microbenchmark for rbpair16_kernel (one fused ResBlock conv pair, 16-bit operands): synthetic data, HIP-event timing and,
// with -DVITS_PHASE_TIMING, per-block phase stamps (DMA fill | conv1 | t tile | conv2 | epilogue) and per-CU residency. Not part of
// the product. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKT_=11 -DDIL_=1 -DC_=128 [-DVITS_PHASE_TIMING] tools/rb16_micro.hip -o /tmp/rb16_micro
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cstdio>
#include <map>
#include <vector>
#ifndef KT_
#define KT_ 11
#endif
#ifndef DIL_
#define DIL_ 1
#endif
#ifndef C_
#define C_ 128
#endif
#ifndef T_
#define T_ 14467  // frames x 64 of the C = 128 stage of the benchmark batch, per utterance
#endif
#ifndef B_
#define B_ 64
#endif
#include "../vits.cpp_amd/csrc/rbpair16.hip"
using namespace vits;

int main() {
    const int C = C_, K = KT_, T = T_, B = B_;
    const int ts = (T + 31) / 32 * 32;
    // weights: A fragments, any finite fp16 bit patterns (timing only)
    const size_t wn = (size_t)C * C * K;
    std::vector<uint16_t> w(wn);
    for (size_t i = 0; i < wn; ++i) w[i] = 0x2000 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff);  // ~0.01
    std::vector<float> bias(C, 0.01f);
    uint16_t *dw1, *dw2, *dx, *dy16;
    float *db, *dyg, *dres;
    hipMalloc(&dw1, wn * 2);
    hipMalloc(&dw2, wn * 2);
    hipMemcpy(dw1, w.data(), wn * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw2, w.data(), wn * 2, hipMemcpyHostToDevice);
    hipMalloc(&db, C * 4);
    hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice);
    const size_t n = (size_t)B * C * ts;
    hipMalloc(&dx, n * 2);
    hipMalloc(&dy16, n * 2);
    hipMalloc(&dyg, n * 4);
    hipMalloc(&dres, n * 4);
    std::vector<uint16_t> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = 0x3000 + (uint16_t)((i * 2246822519u) >> 20 & 0x3ff);
    hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(dres, 0, n * 4);
    PackedConv c1, c2;
    c1.cin = c1.cout = c2.cin = c2.cout = C;
    c1.kt = c2.kt = K;
    c1.wp16 = dw1;
    c2.wp16 = dw2;
    c1.bias = c2.bias = db;
    RbPair16Call f;
    f.x.p = dx;
    f.x.ts = ts;
    f.x.bs = (int64_t)C * ts;
    f.batch = B;
    f.tmax = T;
    f.dil = DIL_;
    f.yg = dyg;
    f.resg = dres;
    f.g_bs = (int64_t)C * ts;
    f.g_ts = ts;
    f.y16.p = dy16;
    f.y16.ts = ts;
    f.y16.bs = (int64_t)C * ts;
    f.y16_slope = 0.1f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_rbpair16(c1, c2, f, VITS_ARITH_F16, nullptr);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_rbpair16(c1, c2, f, VITS_ARITH_F16, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double fl = 2.0 * 2.0 * C * C * K * (double)B * T;
    const double bytes = 12.0 * C * (double)B * T;
#ifdef VITS_PHASE_TIMING
    {
        std::vector<unsigned long long> ph(16 * 65536);
        hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_rb_phase), ph.size() * 8);
        double d[5] = {0, 0, 0, 0, 0}, cyc = 0;
        size_t cnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> percu;
        for (size_t i = 0; i < 65536; ++i) {
            const unsigned long long* q = &ph[16 * i];
            if (!q[0] || !q[5] || q[5] < q[0]) continue;
            for (int k = 0; k < 5; ++k) d[k] += (double)(q[k + 1] - q[k]);
            cyc += (double)(q[10] - q[9]);
            tmin = std::min(tmin, q[0]);
            tmax = std::max(tmax, q[5]);
            ++cnt;
            const unsigned key = (((unsigned)q[8] & 0xf) << 16) | (((unsigned)q[7] >> 8) & 0xff);
            percu[key].push_back({q[0], +1});
            percu[key].push_back({q[5], -1});
        }
        double tres[8] = {0}, tt = 0, tc = 0;
        for (auto& kv : percu) {
            auto& ev = kv.second;
            std::sort(ev.begin(), ev.end());
            int c = 0;
            for (size_t e = 0; e + 1 < ev.size(); ++e) {
                c += ev[e].second;
                const double dt = (double)(ev[e + 1].first - ev[e].first);
                tres[std::min(c, 7)] += dt;
                tt += dt;
                tc += c * dt;
            }
        }
        printf("phases over %zu blocks, us: fill %.2f | conv1 %.2f | t tile %.2f | conv2 %.2f | epilogue %.2f ; block life %.2f us; launch span %.1f us; shader clock during conv1 %.3f GHz\n", cnt,
               d[0] / cnt / 100, d[1] / cnt / 100, d[2] / cnt / 100, d[3] / cnt / 100, d[4] / cnt / 100, (d[0] + d[1] + d[2] + d[3] + d[4]) / cnt / 100, (tmax - tmin) / 100.0,
               cyc / (d[1] * 10.0));
        printf("per-CU residency over %zu CUs: ", percu.size());
        for (int c = 0; c < 8; ++c)
            if (tres[c] > 0) printf("%d blocks %.1f %% | ", c, 100 * tres[c] / tt);
        printf("mean %.2f\n", tc / tt);
    }
#endif
    printf("rbpair16 C=%d k=%d d=%d T=%d B=%d: %.3f ms  %.0f TFLOP/s (algorithmic, both convs)  %.2f TB/s (12 B per element)  (%s)\n", C, K, DIL_, T, B, ms, fl / ms / 1e9,
           bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
