// asan_driver.cpp — runs the oracle under AddressSanitizer + UBSan (the reference's Debug configuration is an ASan build,
// /root/reference/CMakeLists.txt:9-13; the GPU pool has no device-side sanitizer, so the CPU restatement is what gets it).
// Loads a model file, runs one utterance in every semantics mode and arithmetic mode, exercises the operator-level entry
// points and the helper-op restatements (including the index -1 wrap of Q4, which the REFERENCE performs as an out-of-bounds
// write, ggml-util.h:235: the restatement must reproduce its effect without touching memory outside the tensor).
// Test infrastructure only (tests/test_oracle.py::test_oracle_is_clean_under_asan_and_ubsan).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "vits_oracle.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    vo_model* m = vo_load(data.data(), data.size());
    if (!m) {
        std::fprintf(stderr, "load failed: %s\n", vo_last_error());
        return 3;
    }
    int32_t ids[23];
    for (int i = 0; i < 23; ++i) ids[i] = (i & 1) ? 1 + (i * 7) % 30 : 0;
    double sum = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int arith = 0; arith < 3; ++arith)
            for (int n : {1, 3, 23}) {
                vo_opts o{};
                o.mode = mode;
                o.noise_kind = VO_NOISE_COUNTER;
                o.noise_seed = 5 + n;
                o.threads = 3;
                o.arith = arith;
                o.arith_scope = (n == 3) ? VO_SCOPE_ALL_CONVS : VO_SCOPE_FLOW_VOCODER;
                vo_run* r = vo_process_ids(m, ids, n, &o);
                if (!r) {
                    std::fprintf(stderr, "process failed: %s\n", vo_last_error());
                    return 4;
                }
                const int64_t cnt = vo_run_tap(r, "waveform", nullptr, 0);
                std::vector<float> w((size_t)cnt);
                vo_run_tap(r, "waveform", w.data(), w.size());
                for (float v : w) sum += v;
                vo_run_free(r);
            }
    // Q6 (latents outside the spline interval: the masked get / set walk of vits.cpp:832-849) and the emulated ggml tables (Q8), stage one
    // only, explicit noise large enough to leave [-5, 5] — first / last token outside included
    {
        float nd[2 * 23];
        for (int i = 0; i < 2 * 23; ++i) nd[i] = ((i * 37) % 11 - 5) * 1.9f;
        nd[0] = nd[23] = 9.f;
        nd[22] = nd[45] = -9.f;
        for (int mode = 0; mode < 2; ++mode)
            for (int tables = 0; tables < 2; ++tables)
                for (int n : {1, 2, 23}) {
                    vo_opts o{};
                    o.mode = mode;
                    o.noise_kind = VO_NOISE_EXPLICIT;
                    o.noise_dur = nd;
                    o.threads = 2;
                    o.ggml_tables = tables;
                    float logw[23], dur[23];
                    float nd_n[2 * 23];
                    for (int c = 0; c < 2; ++c)
                        for (int t = 0; t < n; ++t) nd_n[c * n + t] = nd[c * 23 + t];
                    o.noise_dur = nd_n;
                    if (vo_log_durations(m, ids, n, &o, logw, dur) != 0) {
                        std::fprintf(stderr, "log_durations failed: %s\n", vo_last_error());
                        return 5;
                    }
                    for (int t = 0; t < n; ++t) sum += logw[t] + dur[t];
                    sum += (double)vo_outside_latents();
                }
        const float tt[6] = {1, 2, 3, 4, 5, 6}, mk[6] = {0, 1, 0, 1, 1, 0};
        float o6[6];
        vo_masked_get(tt, mk, 6, o6);
        sum += o6[1];
    }
    // reference noise stream + a truncated file + helper ops with the -1 wrap
    vo_reference_noise_seed(1);
    float nz[8];
    vo_reference_noise_draw(nz, 8);
    vo_model* bad = vo_load(data.data(), data.size() / 2);
    if (bad) vo_free(bad);
    const int64_t ne[3] = {4, 3, 1};
    float t[12] = {0};
    vo_index_put_last_dim(t, ne, -1, 7.f);
    vo_index_add_last_dim(t, ne, -1, 1.f);
    vo_index_put_last_dim(t, ne, 0, 2.f);
    int32_t tok[64];
    vo_tokenize(m, "hello world", tok, 64);
    vo_free(m);
    std::printf("asan driver ok %.6f %.3f %.1f\n", sum, nz[0], t[3]);
    return 0;
}
