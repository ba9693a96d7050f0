// Developer microbenchmark for wavenet16_kernel (one WaveNet layer of the flow, 16-bit operands) with per-block phase stamps.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITS_PHASE_TIMING tools/wn16_micro.hip -o tools/bin/wn16_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../vits.cpp_amd/csrc/wavenet32.hip"
using namespace vits;
#ifndef L_
#define L_ 225
#endif
#ifndef B_
#define B_ 64
#endif
int main() {
    const int H = 192, K = 5, L = L_, B = B_, ls = (L + 31) / 32 * 32;
    std::vector<uint16_t> w1((size_t)2 * H * H * K), w2((size_t)2 * H * H);
    for (size_t i = 0; i < w1.size(); ++i) w1[i] = 0x2000 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff);
    for (size_t i = 0; i < w2.size(); ++i) w2[i] = 0x2000 + (uint16_t)((i * 2246822519u) >> 20 & 0x3ff);
    std::vector<float> bias(2 * H, 0.01f);
    uint16_t *dw1, *dw2;
    float *db, *dh, *dho, *dout, *dwf;
    hipMalloc(&dw1, w1.size() * 2); hipMemcpy(dw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dw2, w2.size() * 2); hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&db, 2 * H * 4); hipMemcpy(db, bias.data(), 2 * H * 4, hipMemcpyHostToDevice);
    const size_t n = (size_t)B * H * ls;
    hipMalloc(&dh, n * 4); hipMalloc(&dho, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dwf, 16);
    std::vector<float> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(dh, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, n * 4);
    PackedConv in, rs;
    in.cin = H; in.cout = 2 * H; in.kt = K; in.epi = EPI_GATE; in.wp = dwf; in.wp16 = dw1; in.bias = db;
    rs.cin = H; rs.cout = 2 * H; rs.kt = 1; rs.epi = EPI_STD; rs.wp = dwf; rs.wp16 = dw2; rs.bias = db;
    WaveNet32Call c;
    c.h.p = dh; c.h.cs = ls; c.h.bs = (int64_t)H * ls;
    c.h_out.p = dho; c.h_out.cs = ls; c.h_out.bs = (int64_t)H * ls;
    c.outputs.p = dout; c.outputs.cs = ls; c.outputs.bs = (int64_t)H * ls;
    c.batch = B; c.tmax = L; c.hidden = H; c.dil = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_wavenet16(in, rs, c, VITS_ARITH_F16, nullptr);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_wavenet16(in, rs, c, VITS_ARITH_F16, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
#ifdef VITS_PHASE_TIMING
    std::vector<unsigned long long> ph(8 * 65536);
    hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_wn_phase), ph.size() * 8);
    double d[5] = {0, 0, 0, 0, 0}; size_t cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t i = 0; i < 65536; ++i) {
        const unsigned long long* q = &ph[8 * i];
        if (!q[0] || !q[5] || q[5] < q[0]) continue;
        for (int k = 0; k < 5; ++k) d[k] += (double)(q[k + 1] - q[k]);
        tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[5]); ++cnt;
    }
    {
        double a = 0, b2 = 0, c2 = 0; size_t n2 = 0;
        for (size_t i = 0; i < 65536; ++i) {
            const unsigned long long* q = &ph[8 * i];
            if (!q[0] || !q[5] || q[5] < q[0]) continue;
            a += (double)(q[6] - q[2]); b2 += (double)(q[7] - q[6]); c2 += (double)(q[3] - q[7]); ++n2;
        }
        printf("gate phase split, us: barrier after conv %.2f | gate + acts write %.2f | barrier %.2f\n", a / n2 / 100, b2 / n2 / 100, c2 / n2 / 100);
    }
    printf("phases over %zu blocks, us: h tile %.2f | gated conv %.2f | gate + acts %.2f | 1x1 conv %.2f | epilogue %.2f ; span %.1f us\n", cnt, d[0] / cnt / 100, d[1] / cnt / 100,
           d[2] / cnt / 100, d[3] / cnt / 100, d[4] / cnt / 100, (tmax - tmin) / 100.0);
#endif
    printf("wavenet16 H=%d L=%d B=%d: %.1f us per layer (%s)\n", H, L, B, ms * 1000, hipGetErrorString(hipGetLastError()));
    return 0;
}
