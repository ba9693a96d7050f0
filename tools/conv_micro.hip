// Developer microbenchmark for conv_mfma_kernel (one instantiation, synthetic data, HIP-event timing). Not part of the
// product. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKT_=11 -DDIL_=1 -DCIN_=128 [-DVAR_...] tools/conv_micro.hip -o /tmp/conv_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>
#include <array>
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
#ifndef PRE_
#define PRE_ 1
#endif
#ifndef CIN_
#define CIN_ 128
#endif
#ifndef COUT_
#define COUT_ CIN_
#endif
#ifndef T_
#define T_ 14400
#endif
#ifndef B_
#define B_ 64
#endif
int main() {
    const int C = CIN_, CO = COUT_, K = KT_, T = T_, B = B_;
    std::vector<float> w((size_t)C * CO * K, 0.01f), bias(CO, 0.1f);
    PackedConv pc; pc.cin = C; pc.cout = CO; pc.kt = K; pc.epi = EPI_STD;
    auto packed = pack_conv_weights(w.data(), CO, C, K, EPI_STD, 0, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
    float *dw, *db, *dx, *dy;
    hipMalloc(&dw, packed.size() * 4); hipMemcpy(dw, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&db, CO * 4); hipMemcpy(db, bias.data(), CO * 4, hipMemcpyHostToDevice);
    size_t n = (size_t)B * C * T;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, (size_t)B * CO * T * 4);
    std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    pc.wp = dw; pc.bias = db;
    ConvCall c; c.x.p = dx; c.x.cs = T; c.x.bs = (int64_t)C * T; c.y.p = dy; c.y.cs = T; c.y.bs = (int64_t)CO * T;
#ifdef WITH_RES
    c.res = c.x;
#endif
#ifdef TILE_
    c.tile = TILE_;
#endif
    c.batch = B; c.t_in = c.t_out = T; c.dil = DIL_; c.pad_l = (K - 1) * DIL_ / 2; c.pre_act = PRE_; c.slope = 0.1f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_conv(pc, c, nullptr);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_conv(pc, c, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double fl = 2.0 * C * CO * K * (double)B * T;
    {
        int nb_occ = -1;
#if CIN_ <= 32
        auto kfn = conv_mfma_kernel<KT_, DIL_, false, 1, 4, 1, 1, EPI_STD>;
        const size_t lds = 32 * (128 + (KT_ - 1) * DIL_) * 4;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_occ, kfn, 256, lds);
#else
        auto kfn = conv_mfma_kernel<KT_, DIL_, true, 2, 2, 2, 2, EPI_STD>;
        const size_t lds = 2 * 32 * ((128 + (KT_ - 1) * DIL_ + 3 + 63) / 64 * 64) * 4;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_occ, kfn, 320, lds);
#endif
        hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kfn);
        printf("128x128 DB kernel: occupancy API says %d blocks/CU at %zu B LDS (%s); numRegs %d, sharedSizeBytes %zu, maxThreadsPerBlock %d\n", nb_occ, lds,
               hipGetErrorString(oe), fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    }
#ifdef VITS_PHASE_TIMING
    {
        // per-block phases of the LAST launch: prologue (start -> first tile ready), K loop, epilogue; and how busy the
        // device was: sum of K-loop time over blocks / (launch span x resident block slots)
        std::vector<unsigned long long> ph(8 * 65536);
        hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_phase_buf), ph.size() * 8);
        const size_t nb = 65536;
        const int bm = 0, bn = 0;
        double pro = 0, kl = 0, epi = 0, cyc = 0; unsigned long long tmin = ~0ull, tmax = 0; size_t cnt = 0;
        std::vector<double> kls;
        for (size_t i = 0; i < nb; ++i) {
            const unsigned long long* q = &ph[8 * i];
            if (!q[0] || !q[3] || q[3] < q[0]) continue;
            pro += (double)(q[1] - q[0]); kl += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]);
            kls.push_back((double)(q[2] - q[1]));
            cyc += (double)(q[5] - q[4]);
            tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[3]); ++cnt;
        }
        {
            // per-CU view: HW_ID bits [11:8] CU, [12] SH, [15:13] SE (gfx9), XCC_ID low bits; events (+1 at start, -1 at end)
            std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> percu;
            unsigned long long tmin0 = ~0ull;
            for (size_t i = 0; i < nb; ++i)
                if (ph[8 * i] && ph[8 * i + 3]) tmin0 = std::min(tmin0, ph[8 * i]);
            for (size_t i = 0; i < nb; ++i) {
                const unsigned long long* q = &ph[8 * i];
                if (!q[0] || !q[3] || q[3] < q[0]) continue;
                const unsigned hw = (unsigned)q[6], xcc = (unsigned)q[7] & 0xf;
                const unsigned key = (xcc << 16) | ((hw >> 8) & 0xff);
                percu[key].push_back({q[0], +1});
                percu[key].push_back({q[3], -1});
            }
            double t1 = 0, t2 = 0, t3 = 0, t0 = 0, tc = 0; size_t maxc = 0;
            for (auto& kv : percu) {
                auto& ev = kv.second;
                std::sort(ev.begin(), ev.end());
                int c = 0;
                for (size_t e = 0; e + 1 < ev.size(); ++e) {
                    c += ev[e].second;
                    const double dt = (double)(ev[e + 1].first - ev[e].first);
                    (c == 0 ? t0 : c == 1 ? t1 : c == 2 ? t2 : t3) += dt; tc += c * dt;
                    maxc = std::max(maxc, (size_t)c);
                }
            }
            {
                // timeline of one CU in the middle of the run: block start / K-loop start / K-loop end / end, us from the first start
                std::vector<std::array<double, 4>> tl;
                const unsigned want = percu.begin()->first;
                for (size_t i = 0; i < nb; ++i) {
                    const unsigned long long* q = &ph[8 * i];
                    if (!q[0] || !q[3] || q[3] < q[0]) continue;
                    const unsigned hw = (unsigned)q[6], xcc = (unsigned)q[7] & 0xf;
                    if (((xcc << 16) | ((hw >> 8) & 0xff)) != want) continue;
                    tl.push_back({(q[0] - tmin0) / 100.0, (q[1] - tmin0) / 100.0, (q[2] - tmin0) / 100.0, (q[3] - tmin0) / 100.0});
                }
                {
                    std::vector<unsigned long long> we(8 * 65536);
                    hipMemcpyFromSymbol(we.data(), HIP_SYMBOL(vits_wave_end), we.size() * 8);
                    std::vector<unsigned long long> ws(8 * 65536);
                    hipMemcpyFromSymbol(ws.data(), HIP_SYMBOL(vits_wave_start), ws.size() * 8);
                    std::vector<std::array<double, 9>> tws;
                    for (size_t i = 0; i < nb; ++i) {
                        const unsigned long long* q = &ph[8 * i];
                        if (!q[0] || !q[3] || q[3] < q[0]) continue;
                        const unsigned hw = (unsigned)q[6], xcc = (unsigned)q[7] & 0xf;
                        if (((xcc << 16) | ((hw >> 8) & 0xff)) != want) continue;
                        std::array<double, 9> a{};
                        for (int w = 0; w < 5; ++w) a[w] = ws[8 * i + w] ? ((double)ws[8 * i + w] - (double)tmin0) / 100.0 : -1.0;
                        // HW_ID: [3:0] wave slot, [5:4] SIMD
                        for (int k = 0; k < 3; ++k) a[5 + k] = (double)((ws[8 * i + 5 + k] >> 4) & 3);
                        a[8] = (q[2] - tmin0) / 100.0;
                        tws.push_back(a);
                    }
                    std::sort(tws.begin(), tws.end());
                    for (size_t i = 12; i < tws.size() && i < 18; ++i)
                        printf("   cu0 block %2zu: waves start %8.2f %8.2f %8.2f %8.2f producer %8.2f | SIMD of wave 0 / 1 / producer: %.0f %.0f %.0f | kloop-end %8.2f\n", i, tws[i][0], tws[i][1], tws[i][2], tws[i][3], tws[i][4], tws[i][5], tws[i][6], tws[i][7], tws[i][8]);
                    std::vector<std::array<double, 7>> tw;
                    for (size_t i = 0; i < nb; ++i) {
                        const unsigned long long* q = &ph[8 * i];
                        if (!q[0] || !q[3] || q[3] < q[0]) continue;
                        const unsigned hw = (unsigned)q[6], xcc = (unsigned)q[7] & 0xf;
                        if (((xcc << 16) | ((hw >> 8) & 0xff)) != want) continue;
                        std::array<double, 7> a{(q[0] - tmin0) / 100.0, (q[2] - tmin0) / 100.0, 0, 0, 0, 0, 0};
                        for (int w = 0; w < 5; ++w) a[2 + w] = we[8 * i + w] ? ((double)we[8 * i + w] - (double)tmin0) / 100.0 : -1.0;
                        tw.push_back(a);
                    }
                    std::sort(tw.begin(), tw.end());
                    for (size_t i = 12; i < tw.size() && i < 18; ++i)
                        printf("   cu0 block %2zu: start %8.2f kloop-end %8.2f | waves retire: %8.2f %8.2f %8.2f %8.2f producer %8.2f\n", i, tw[i][0], tw[i][1], tw[i][2], tw[i][3], tw[i][4], tw[i][5], tw[i][6]);
                }
                std::sort(tl.begin(), tl.end());
                for (size_t i = 8; i < tl.size() && i < 20; ++i) printf("   cu0 block %2zu: start %8.2f  kloop %8.2f .. %8.2f  end %8.2f\n", i, tl[i][0], tl[i][1], tl[i][2], tl[i][3]);
            }
            const double tt = t0 + t1 + t2 + t3;
            printf("per-CU residency over %zu distinct CUs: 0 blocks %.1f %%, 1 block %.1f %%, 2 blocks %.1f %%, >=3 %.1f %% of the time (max %zu, mean %.2f)\n", percu.size(),
                   100 * t0 / tt, 100 * t1 / tt, 100 * t2 / tt, 100 * t3 / tt, maxc, tc / tt);
        }
        {
            std::vector<unsigned long long> cb(8 * 65536);
            hipMemcpyFromSymbol(cb.data(), HIP_SYMBOL(vits_chunk_buf), cb.size() * 8);
            double comp[4] = {0, 0, 0, 0}, bar[4] = {0, 0, 0, 0}; size_t n = 0;
            for (size_t i = 0; i < nb; ++i) {
                const unsigned long long* q = &ph[8 * i];
                if (!q[0] || !q[3] || q[3] < q[0]) continue;
                unsigned long long prev = q[4];  // K loop start (shader clock)
                for (int c = 0; c < 4; ++c) {
                    comp[c] += (double)(cb[8 * i + 2 * c] - prev);
                    bar[c] += (double)(cb[8 * i + 2 * c + 1] - cb[8 * i + 2 * c]);
                    prev = cb[8 * i + 2 * c + 1];
                }
                ++n;
            }
            printf("per chunk (cycles): ");
            for (int c = 0; c < 4; ++c) printf(" c%d compute %.0f barrier-wait %.0f |", c, comp[c] / n, bar[c] / n);
            printf(" ideal compute %d\n", KT_ * 16 * 4 * 64);
        }
        {
            std::vector<unsigned long long> pb(4 * 65536);
            hipMemcpyFromSymbol(pb.data(), HIP_SYMBOL(vits_prod_buf), pb.size() * 8);
            double s4[4] = {0, 0, 0, 0}; size_t n = 0;
            for (size_t i = 0; i < nb; ++i) {
                if (!pb[4 * i + 3]) continue;
                for (int k = 0; k < 4; ++k) s4[k] += (double)pb[4 * i + k];
                ++n;
            }
            const int nch = (CIN_ + 31) / 32 - 1;
            if (n) printf("producer per chunk (cycles, 3-buffer path): issue %.0f | DMA wait %.0f | post-process %.0f | barrier %.0f\n", s4[0] / n / nch, s4[1] / n / nch, s4[2] / n / nch, s4[3] / n / nch);
        }
        {
            std::vector<unsigned long long> eb(8 * 65536);
            hipMemcpyFromSymbol(eb.data(), HIP_SYMBOL(vits_epi_buf), eb.size() * 8);
            double d[6] = {0, 0, 0, 0, 0, 0}; size_t n = 0;
            for (size_t i = 0; i < nb; ++i) {
                const unsigned long long* q = &eb[8 * i];
                if (!q[0] || !q[6] || !q[2]) continue;  // (only blocks that took the wide path stamp 1..5)
                for (int k = 0; k < 6; ++k) d[k] += (double)(q[k + 1] - q[k]);
                ++n;
            }
            if (n) printf("epilogue of wave 0 (cycles, %zu wide blocks): entry->setup %.0f | setup->sub-tile 0 %.0f | sub-tile 0 %.0f | 1 %.0f | 2 %.0f | 3 + exit %.0f\n", n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n);
        }
        std::sort(kls.begin(), kls.end());
        printf("phases over %zu blocks (tile %dx%d), 10 ns ticks -> us: prologue %.2f  k-loop %.2f (p10 %.2f p90 %.2f)  epilogue %.2f; span %.1f us; blocks*life/span = %.1f resident; shader clock in the K loop %.3f GHz, MFMA issue efficiency of the K loop %.3f\n",
               cnt, bm, bn, pro / cnt / 100, kl / cnt / 100, kls[kls.size() / 10] / 100, kls[kls.size() * 9 / 10] / 100, epi / cnt / 100, (tmax - tmin) / 100.0,
               (pro + kl + epi) / (double)(tmax - tmin), cyc / (kl * 10.0), (double)cnt * ((CIN_ + 31) / 32) * KT_ * 16 * 4 * 64.0 / cyc);
    }
#endif
    printf("C=%d k=%d d=%d T=%d B=%d: %.3f ms  %.1f TFLOP/s  (%s)\n", C, K, DIL_, T, B, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
