// Developer microbenchmark for dds_layer_kernel (misc_kernels.hip) at batch-1 size: HIP-event time per launch with cold caches
// (a 1 GiB fill between launches) and per-phase timestamps of every block. Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITS_PHASE_TIMING tools/dds_micro.hip vits.cpp_amd/csrc/conv_mfma*.o ... (see tools/jobs/ddsmicro.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../vits.cpp_amd/csrc/misc_kernels.hip"
using namespace vits;
// fragment order of conv_mfma.hip pack_conv_weights for a 1x1 conv (EPI_STD)
static std::vector<float> pack1x1(const float* w, int cout, int cin, int* mtiles, int* nchunks) {
    *mtiles = (cout + 31) / 32;
    *nchunks = (cin + 31) / 32;
    const int mt_pad = (*mtiles + 3) / 4 * 4;
    std::vector<float> out((size_t)mt_pad * *nchunks * 16 * 64, 0.f);
    for (int mt = 0; mt < *mtiles; ++mt)
        for (int c = 0; c < *nchunks; ++c)
            for (int pr = 0; pr < 16; ++pr)
                for (int l = 0; l < 64; ++l) {
                    const int ci = c * 32 + 2 * pr + (l >> 5), co = mt * 32 + (l & 31);
                    if (ci < cin && co < cout) out[((((size_t)mt * *nchunks + c) * 4 + pr / 4) * 64 + l) * 4 + (pr & 3)] = w[(size_t)co * cin + ci];
                }
    return out;
}
int main(int argc, char** argv) {
    const int H = 192, T = argc > 1 ? atoi(argv[1]) : 257, B = argc > 2 ? atoi(argv[2]) : 1, K = 3;
    const int dil = argc > 3 ? atoi(argv[3]) : 9;
    std::vector<float> w((size_t)H * H), v(H, 0.1f), dw((size_t)H * K, 0.3f), one(H, 1.f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 0.1f - 0.05f;
    PackedConv pc; pc.cin = H; pc.cout = H; pc.kt = 1; pc.epi = EPI_STD;
    auto packed = pack1x1(w.data(), H, H, &pc.mtiles_used, &pc.nchunks);
    pc.rows = H; pc.mtiles = (pc.mtiles_used + 3) / 4 * 4;
    float *dwp, *db, *ddw, *dg, *dx, *dy, *big;
    hipMalloc(&dwp, packed.size() * 4); hipMemcpy(dwp, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&db, H * 4); hipMemcpy(db, v.data(), H * 4, hipMemcpyHostToDevice);
    hipMalloc(&dg, H * 4); hipMemcpy(dg, one.data(), H * 4, hipMemcpyHostToDevice);
    hipMalloc(&ddw, H * K * 4); hipMemcpy(ddw, dw.data(), H * K * 4, hipMemcpyHostToDevice);
    const int ts = (T + 63) / 64 * 64;
    size_t n = (size_t)B * H * ts;
    std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    const size_t bigN = 256u << 20; hipMalloc(&big, bigN * 4);
    pc.wp = dwp; pc.bias = db;
    const int nblk = (T + 31) / 32;
    unsigned long long* dbg; hipMalloc(&dbg, nblk * 16 * 8); hipMemset(dbg, 0, nblk * 16 * 8);
    TensorRef x, y; x.p = dx; x.cs = ts; x.bs = (int64_t)H * ts; y = x; y.p = dy;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cold = 0; cold < 2; ++cold) {
        float best = 1e9f, sum = 0;
        const int reps = 10;
        for (int i = 0; i < reps + 1; ++i) {
            if (cold) hipMemsetAsync(big, i, bigN * 4, nullptr);
            g_dds_dbg = i == reps ? dbg : nullptr;
            hipEventRecord(e0, nullptr);
            hipError_t e = launch_dds_layer(x, y, ddw, db, dg, db, pc, dg, db, nullptr, B, H, T, K, dil, 1e-5f, 0, nullptr);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            if (e != hipSuccess) { printf("launch failed %d\n", (int)e); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i > 0 && i < reps) { best = std::min(best, ms); sum += ms; }
        }
        printf("%s: avg %.1f us  best %.1f us per launch (T=%d B=%d dil=%d, %d blocks)\n", cold ? "cold" : "warm", sum / (reps - 1) * 1e3, best * 1e3, T, B, dil, nblk * B);
        std::vector<unsigned long long> h(nblk * 16);
        hipMemcpy(h.data(), dbg, nblk * 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int bl = 0; bl < nblk; ++bl) t0 = std::min(t0, h[bl * 16]);
        for (int bl = 0; bl < nblk; ++bl) {
            printf("  block %d: start +%.2f us | lens %.2f | loads %.2f | dw+LN1 %.2f | mfma %.2f | bar %.2f | store-lds %.2f | LN2+out %.2f\n", bl, (h[bl * 16] - t0) * 0.01,
                   (h[bl * 16 + 1] - h[bl * 16]) * 0.01, (h[bl * 16 + 2] - h[bl * 16 + 1]) * 0.01, (h[bl * 16 + 3] - h[bl * 16 + 2]) * 0.01, (h[bl * 16 + 4] - h[bl * 16 + 3]) * 0.01,
                   (h[bl * 16 + 5] - h[bl * 16 + 4]) * 0.01, (h[bl * 16 + 6] - h[bl * 16 + 5]) * 0.01, (h[bl * 16 + 7] - h[bl * 16 + 6]) * 0.01);
        }
    }
    return 0;
}
