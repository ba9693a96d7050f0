// vits_oracle_exact.cpp — stage one (text encoder + stochastic duration predictor) of the EMULATED-ggml mode (vo_opts.ggml_tables = 1) in the one
// order of operations that include/vits_exact_math.h defines and the product's device code (vits.cpp_amd/csrc/exact_stage1.hip) shares: every
// number below is produced by a vx_* element function called once per output element; this file only walks the graph
// (/root/reference/src/vits.cpp:244-440 text_encoder_graph, :927-972 stochastic_duration_predictor_graph, :646-692 DDS, :855-899 conv flow,
// :804-852 spline step, :901-925 affine, :995-1001 durations) in the same sequence as vits_oracle.cpp's text_encoder() / duration_predictor().
// Compiled with -ffp-contract=off (oracle/Makefile): a * b + c below is two roundings, as in the device build.
// TEST INFRASTRUCTURE like the rest of oracle/. vits_oracle.cpp's own loops (other summation grouping, the C library's exp / log) stay the
// independent restatement; tests/test_oracle.py compares the two at tolerance.
#include "vits_oracle_exact.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <thread>

#include "../include/vits_exact_math.h"

namespace vo_exact {
namespace {

void parallel_for(int threads, int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    if (threads == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + threads - 1) / threads;
    for (int i = 0; i < threads; ++i) {
        const int64_t b = i * per, e = std::min(n, b + per);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto& t : th) t.join();
}

struct Act {
    int C = 0, T = 0;
    std::vector<float> d;
    Act() {}
    Act(int c, int t) : C(c), T(t), d((size_t)c * t, 0.f) {}
};

struct Tables {
    std::vector<uint16_t> gelu, exp;
    Tables() : gelu(65536), exp(65536) { vx_build_ggml_tables(gelu.data(), exp.data()); }
};
const Tables& tables() {
    static const Tables t;
    return t;
}

// y = conv(x) [+ relu] [* post_scale] [+ res]: w [cout][cin][K] (a Conv1d tensor, file ne = [K, cin, cout], or a Linear one, ne = [cin, cout], K = 1)
Act conv(const Act& x, const TensorRef& w, const float* bias, int K, int dil, int pad_l, bool relu, const float* post_scale, const Act* res, int threads) {
    const int cin = x.C, T = x.T;
    const bool linear = w.rank == 2;
    const int cout = linear ? (int)w.ne[1] : (int)w.ne[2];
    if (linear ? ((int)w.ne[0] != cin || K != 1) : ((int)w.ne[1] != cin || (int)w.ne[0] != K)) throw std::runtime_error("exact conv: shape mismatch");
    Act y(cout, T);
    parallel_for(threads, (int64_t)cout * T, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const int co = (int)(i / T), t = (int)(i % T);
            float v = vx_conv_elem(x.d.data(), T, cin, T, w.d + (size_t)co * cin * K, bias ? bias[co] : 0.f, K, dil, pad_l, t);
            if (relu) v = v > 0.f ? v : 0.f;
            if (post_scale) v = v * *post_scale;
            if (res) v = res->d[i] + v;
            y.d[i] = v;
        }
    });
    return y;
}

void layer_norm(Act& x, const TensorRef& g, const TensorRef& b, float eps, bool gelu, int threads) {
    const uint16_t* tab = gelu ? tables().gelu.data() : nullptr;
    parallel_for(threads, x.T, [&](int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; ++t) vx_layer_norm_column(x.d.data() + t, x.T, x.C, g.d, b.d, eps, tab);
    });
}

Act dds(const ModelView& m, const std::string& base, Act x, const Act* g, int threads) {
    const int C = x.C, T = x.T;
    if (g)
        for (size_t i = 0; i < x.d.size(); ++i) x.d[i] = x.d[i] + g->d[i];  // vits.cpp:651-653
    for (int i = 0; i < m.dds_layers; ++i) {
        const std::string si = std::to_string(i);
        const TensorRef wd = m.T(base + "convs_dilated." + si + ".weight"), bd = m.T(base + "convs_dilated." + si + ".bias");
        const int K = m.dp_k;
        int dil = 1;
        for (int e = 0; e < i; ++e) dil *= K;     // :659
        const int pad = (K * dil - dil) / 2;     // :660
        Act h(C, T);
        parallel_for(threads, (int64_t)C * T, [&](int64_t b, int64_t e) {
            for (int64_t idx = b; idx < e; ++idx) {
                const int ch = (int)(idx / T), t = (int)(idx % T);
                h.d[idx] = vx_depthwise_elem(x.d.data() + (size_t)ch * T, T, wd.d + (size_t)ch * K, bd.d[ch], K, dil, pad, t);
            }
        });
        layer_norm(h, m.T(base + "norms_1." + si + ".weight"), m.T(base + "norms_1." + si + ".bias"), 1e-5f, true, threads);  // :668-673
        const TensorRef wp = m.T(base + "convs_pointwise." + si + ".weight"), bp = m.T(base + "convs_pointwise." + si + ".bias");
        Act p = conv(h, wp, bp.d, 1, 1, 0, false, nullptr, nullptr, threads);
        layer_norm(p, m.T(base + "norms_2." + si + ".weight"), m.T(base + "norms_2." + si + ".bias"), 1e-5f, true, threads);  // :679-687
        for (size_t e = 0; e < x.d.size(); ++e) x.d[e] = x.d[e] + p.d[e];                                                      // :688
    }
    return x;
}

}  // namespace

void stage_one(const ModelView& m, bool refmode, const int32_t* ids, int T, const float* noise, int threads, StageOne& out) {
    const int H = m.hidden, hd = H / m.heads, F = m.flow_size;
    const Tables& tab = tables();
    // ---- text encoder (vits.cpp:244-440) ----
    const TensorRef emb = m.T("text_encoder.embed_tokens.weight");
    const int vocab = (int)emb.ne[1];
    Act x(H, T);
    const float sc = (float)std::sqrt((double)H);  // :263
    for (int t = 0; t < T; ++t) {
        const int id = ids[t];
        if (id < 0 || id >= vocab) throw std::runtime_error("token id out of range");
        for (int ch = 0; ch < H; ++ch) x.d[(size_t)ch * T + t] = emb.d[(size_t)id * H + ch] * sc;
    }
    const float scaling = (float)std::pow((double)hd, -0.5);  // :296
    std::vector<float> scratch((size_t)std::max(threads, 1) * T);
    for (int l = 0; l < m.layers; ++l) {
        const std::string base = "text_encoder.encoder.layers." + std::to_string(l) + ".";
        auto lin = [&](const char* name, const float* post_scale, const Act& in, const Act* res) {
            return conv(in, m.T(base + "attention." + name + ".weight"), m.T(base + "attention." + name + ".bias").d, 1, 1, 0, false, post_scale, res, threads);
        };
        Act q = lin("q_proj", &scaling, x, nullptr), k = lin("k_proj", nullptr, x, nullptr), v = lin("v_proj", nullptr, x, nullptr);
        Act att(H, T);
        const TensorRef Ek = m.T(base + "attention.emb_rel_k"), Ev = m.T(base + "attention.emb_rel_v");
        {
            // one (head, query) per work item; a scratch row of T scores per worker
            const int64_t items = (int64_t)m.heads * T;
            const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(threads, items));
            std::vector<std::thread> th;
            const int64_t per = (items + nth - 1) / nth;
            for (int w = 0; w < nth; ++w) {
                const int64_t b = w * per, e = std::min(items, b + per);
                if (b >= e) break;
                th.emplace_back([&, b, e, w] {
                    float* s = scratch.data() + (size_t)w * T;
                    for (int64_t it = b; it < e; ++it) {
                        const int h = (int)(it / T), i = (int)(it % T);
                        const size_t off = (size_t)h * hd * T;
                        vx_attention_query(q.d.data() + off, k.d.data() + off, v.d.data() + off, T, hd, T, m.window, Ek.d, Ev.d, i, s, tab.exp.data(), att.d.data() + off);
                    }
                });
            }
            for (auto& t : th) t.join();
        }
        x = lin("out_proj", nullptr, att, &x);  // :358, :367 (residual + cur)
        layer_norm(x, m.T(base + "layer_norm.weight"), m.T(base + "layer_norm.bias"), m.ln_eps, false, threads);
        const int pl = (m.ffn_k - 1) / 2;  // :388 (right pad k / 2: the output keeps the length T)
        Act h1 = conv(x, m.T(base + "feed_forward.conv_1.weight"), m.T(base + "feed_forward.conv_1.bias").d, m.ffn_k, 1, pl, true, nullptr, nullptr, threads);
        x = conv(h1, m.T(base + "feed_forward.conv_2.weight"), m.T(base + "feed_forward.conv_2.bias").d, m.ffn_k, 1, pl, false, nullptr, &x, threads);  // :416
        layer_norm(x, m.T(base + "final_layer_norm.weight"), m.T(base + "final_layer_norm.bias"), m.ln_eps, false, threads);
    }
    out.enc = x.d;
    out.stats = conv(x, m.T("text_encoder.project.weight"), m.T("text_encoder.project.bias").d, 1, 1, 0, false, nullptr, nullptr, threads).d;  // :429
    (void)F;

    // ---- stochastic duration predictor, reverse (vits.cpp:927-972) ----
    const std::string dp = "duration_predictor.";
    Act c0 = conv(x, m.T(dp + "conv_pre.weight"), m.T(dp + "conv_pre.bias").d, 1, 1, 0, false, nullptr, nullptr, threads);  // :934
    c0 = dds(m, dp + "conv_dds.", c0, nullptr, threads);                                                                       // :941
    Act cond = conv(c0, m.T(dp + "conv_proj.weight"), m.T(dp + "conv_proj.bias").d, 1, 1, 0, false, nullptr, nullptr, threads);  // :943
    Act z(2, T);
    for (int i = 0; i < 2 * T; ++i) z.d[i] = noise[i] * m.noise_scale_dur;  // :948-949
    const int nb = m.dp_bins;
    if (nb > VX_MAX_BINS) throw std::runtime_error("exact stage one: more than 16 spline bins");
    const float B = m.dp_tail;
    const float inv_sqrt = (float)(1.0 / std::sqrt((double)H));                        // :877
    const float constant = (float)std::log(std::exp(1.0 - (double)1e-3f) - 1.0);       // :826
    out.outside_latents = 0;
    for (int f = m.dp_flows; f > -1; --f) {  // :953-965
        if (f == 1) continue;
        for (int t = 0; t < T; ++t) std::swap(z.d[t], z.d[(size_t)T + t]);  // flip :956
        const std::string fb = dp + "flows." + std::to_string(f) + ".";
        if (f == 0) {
            const TensorRef tr = m.T(fb + "translate"), ls = m.T(fb + "log_scale");
            for (int ch = 0; ch < 2; ++ch) {
                const float e = std::exp(refmode ? ls.d[ch] : -ls.d[ch]);  // Q5 (:913-918); two values per model, computed on the host on both sides
                for (int t = 0; t < T; ++t) z.d[(size_t)ch * T + t] = (z.d[(size_t)ch * T + t] - tr.d[ch]) * e;
            }
            continue;
        }
        Act z0(1, T);
        std::copy(z.d.begin(), z.d.begin() + T, z0.d.begin());
        Act h = conv(z0, m.T(fb + "conv_pre.weight"), m.T(fb + "conv_pre.bias").d, 1, 1, 0, false, nullptr, nullptr, threads);  // :864
        h = dds(m, fb + "conv_dds.", h, &cond, threads);                                                                          // :868
        Act u = conv(h, m.T(fb + "conv_proj.weight"), m.T(fb + "conv_proj.bias").d, 1, 1, 0, false, nullptr, nullptr, threads);  // :871 [3 nb - 1][T]
        float* x1 = z.d.data() + T;
        std::vector<float> res(T), inside(T);
        for (int t = 0; t < T; ++t) {
            inside[t] = (x1[t] >= -B && x1[t] <= B) ? 1.f : 0.f;  // :819-823
            if (inside[t] != 1.f) ++out.outside_latents;
        }
        parallel_for(threads, T, [&](int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                const bool in = inside[t] == 1.f;
                // reference mode (Q6 :832-840): an outside token goes through the spline as input 0 with zeroed parameters; HF: identity outside
                if (!refmode && !in) res[t] = x1[t];
                else
                    res[t] = vx_spline_token(in ? x1[t] : 0.f, u.d.data() + t, T, nb, B, inv_sqrt, constant, refmode ? 1 : 0, t == T - 1, !in, tab.exp.data());
            }
        });
        if (!refmode) {
            std::copy(res.begin(), res.end(), x1);
        } else {
            // :832 outputs = masked_set(zeros, outside, masked_get(x, inside)); :849 outputs = masked_set(outputs, inside, result) — sequential walks
            std::vector<float> outv(T, 0.f);
            int index = 0;
            for (int t = 0; t < T; ++t)
                if (inside[t] != 1.f) {
                    const int s = index++;
                    outv[t] = inside[s] == 1.f ? x1[s] : 0.f;
                }
            index = 0;
            for (int t = 0; t < T; ++t)
                if (inside[t] == 1.f) outv[t] = res[index++];
            std::copy(outv.begin(), outv.end(), x1);
        }
    }
    out.logw.assign(z.d.begin(), z.d.begin() + T);  // :967-968
    out.dur.resize(T);
    const float length_scale = (float)(1.0 / m.speaking_rate);
    for (int t = 0; t < T; ++t) out.dur[t] = vx_duration(out.logw[t], length_scale);  // :995-1001
}

}  // namespace vo_exact
