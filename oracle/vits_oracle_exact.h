// vits_oracle_exact.h — internal interface between vits_oracle.cpp and vits_oracle_exact.cpp (the exact-order stage one of the emulated-ggml mode).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace vo_exact {

struct TensorRef {
    const float* d;
    int rank;
    int64_t ne[4];  // file order (ggml ne)
};
struct ModelView {
    std::function<TensorRef(const std::string&)> T;  // throws std::runtime_error when the tensor is missing
    int hidden, layers, heads, window, ffn_k, flow_size, dp_k, dds_layers, dp_bins, dp_flows;
    float ln_eps, dp_tail, noise_scale_dur, speaking_rate;
};
struct StageOne {
    std::vector<float> enc;    // [H][T]
    std::vector<float> stats;  // [2F][T]: prior means, then prior log-variances
    std::vector<float> logw;   // [T]
    std::vector<float> dur;    // [T] integer-valued
    int64_t outside_latents = 0;
};
// refmode: VO_MODE_REFERENCE semantics (Q3 / Q4 / Q5 / Q6); noise [2][T] (unscaled N(0,1) draws); threads > 0
void stage_one(const ModelView& m, bool refmode, const int32_t* ids, int T, const float* noise, int threads, StageOne& out);

}  // namespace vo_exact
