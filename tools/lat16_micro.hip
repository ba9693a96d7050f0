// Developer microbenchmark for conv_lat16_kernel (phase stamps). Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITS_PHASE_TIMING
//   -DCIN_=768 -DCOUT_=192 -DKT_=3 -DT_=128 tools/lat16_micro.hip -o tools/bin/lat16_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define VITS_MICRO_KT 3
#define VITS_MICRO_DIL 1
#include "../vits.cpp_amd/csrc/conv_mfma.hip"
using namespace vits;
#ifndef CIN_
#define CIN_ 768
#endif
#ifndef COUT_
#define COUT_ 192
#endif
#ifndef KT_
#define KT_ 3
#endif
#ifndef T_
#define T_ 128
#endif
int main() {
    const int C = CIN_, CO = COUT_, K = KT_, T = T_, TS = (T + 3) / 4 * 4;
    std::vector<float> w((size_t)C * CO * K), bias(CO, 0.1f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    PackedConv pc; pc.cin = C; pc.cout = CO; pc.kt = K; pc.epi = EPI_STD;
    auto packed = pack_conv_weights(w.data(), CO, C, K, EPI_STD, 0, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
    auto pl = repack_conv_weights_l16(packed, pc.mtiles, pc.nchunks, K);
    float *dw, *dl, *db, *dx, *dy;
    hipMalloc(&dw, packed.size() * 4); hipMemcpy(dw, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dl, pl.size() * 4); hipMemcpy(dl, pl.data(), pl.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&db, CO * 4); hipMemcpy(db, bias.data(), CO * 4, hipMemcpyHostToDevice);
    std::vector<float> hx((size_t)C * TS); for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u) >> 4 & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&dx, hx.size() * 4); hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dy, (size_t)CO * TS * 4);
    pc.wp = dw; pc.wp_l16 = dl; pc.bias = db;
    ConvCall c; c.x.p = dx; c.x.cs = TS; c.x.bs = (int64_t)C * TS; c.y.p = dy; c.y.cs = TS; c.y.bs = (int64_t)CO * TS;
    c.batch = 1; c.t_in = c.t_out = T; c.dil = 1; c.pad_l = (K - 1) / 2; c.pre_act = 0; c.post_act = 1;
    // a big buffer written between launches so that the weights are not in L2 when the launch starts (as in a real step)
    float* trash; const size_t tn = 64u << 20; hipMalloc(&trash, tn * 4);
    for (int tile : {(int)TILE_LAT16, (int)TILE_NARROW}) {
        c.tile = tile;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 6; ++i) {
            if (!getenv("WARM")) hipMemsetAsync(trash, i, tn * 4, nullptr);
            hipEventRecord(e0, nullptr);
            hipError_t e = launch_conv(pc, c, nullptr);
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i) { best = std::min(best, ms); sum += ms; }
            if (e != hipSuccess) printf("launch error %s\n", hipGetErrorString(e));
        }
        printf("tile %d: %.1f us best, %.1f us mean (cold weights)\n", tile, best * 1e3, sum / 5 * 1e3);
        if (tile == TILE_LAT16) {
            std::vector<unsigned long long> ph(8 * 65536);
            hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_phase_buf), ph.size() * 8);
            double d[4] = {0, 0, 0, 0}; int n = 0; unsigned long long tmin = ~0ull, tmax = 0;
            for (int i = 0; i < 65536; ++i) {
                const unsigned long long* q = &ph[8 * i];
                if (!q[0] || !q[4]) continue;
                d[0] += q[1] - q[0]; d[1] += q[2] - q[1]; d[2] += q[3] - q[2]; d[3] += q[4] - q[3]; ++n;
                tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[4]);
            }
            if (n) printf("  %d blocks, 10 ns ticks -> us: setup + A prefetch issue %.2f | fill %.2f | barrier %.2f | K loop %.2f ; first start -> last K end %.1f us\n", n, d[0] / n / 100, d[1] / n / 100,
                          d[2] / n / 100, d[3] / n / 100, (tmax - tmin) / 100.0);
        }
    }
    return 0;
}
