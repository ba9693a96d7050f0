// engine_stage1_exact.cpp — stage one of a call in the emulated-ggml mode (vits_model_set_ggml_tables(model, 1)): the text encoder
// (/root/reference/src/vits.cpp:244-440) and the stochastic duration predictor, reverse (:927-972), launched element kernel by element kernel
// (exact_stage1.hip) in the sequence of the test oracle's exact-order translation unit (oracle/, never linked here) — same element functions (include/vits_exact_math.h), same operands, same
// order, fp contraction off on both sides: the log-durations, and so the durations (vits.cpp:996-1001), are bit-identical to the oracle's.
// Works on the tensors as the file holds them (torch layout; uploaded by Engine::set_ggml_tables), fills the same stage-one buffers as the
// throughput path (x = encoder output, stats, z, dur / cum / frames / stage_lens), so everything behind the frame-count read is unchanged.
// ~1 % of the path's work; one thread per output element — this mode is a measurement instrument (how far do ggml's tables move the durations),
// not a serving configuration.
#include <cmath>

#include "engine_internal.h"

namespace vits {

int Engine::run_stage_one_exact(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, id_stride = c.id_stride, ts = c.ts, Tmax = c.Tmax, n_up = c.n_up;
    const bool refmode = c.refmode;
    const int H = hp.hidden, F = hp.flow_size, heads = hp.heads, hd = H / heads;
    const std::vector<int>& tlen = c.tlen;
    Call::S1& s1 = c.s1;
    const int* dl = s1.lens;
    auto TR = make_ref;
    auto sub = sub_rows;
    if (arith_scope == VITS_ARITH_SCOPE_ALL_CONVS && arith != VITS_ARITH_F32) {
        err = "ggml tables mode 1 (exact order) needs fp32 stage-one arithmetic: use VITS_ARITH_SCOPE_FLOW_VOCODER";
        return -1;
    }
    if ((size_t)B * heads * Tmax * ts * sizeof(float) > ((size_t)1 << 31)) {
        err = "ggml tables mode 1 (exact order): batch x ids^2 too large for its score scratch; use a smaller batch";
        return -1;
    }
    auto W = [&](const std::string& name) -> const ExactTensor* {
        auto it = exact_w_.find(name);
        if (it == exact_w_.end()) {
            err = "[ERROR] tensor not found: " + name;
            return nullptr;
        }
        return &it->second;
    };
    const uint16_t* gelu_tab = ggml_tabs_.gelu;
    const uint16_t* exp_tab = ggml_tabs_.exp;
    TensorRef none;
    // y = conv(x) [relu] [* scale] [+ res]; Conv1d tensors have file ne = [K, cin, cout], Linear ones [cin, cout]
    auto conv_op = [&](const char* label, TensorRef x, int cin, const std::string& wname, const std::string& bname, TensorRef y, int K, int pad_l, bool relu, const float* scale,
                       TensorRef res) -> int {
        const ExactTensor* w = W(wname);
        const ExactTensor* b = W(bname);
        if (!w || !b) return -1;
        const bool linear = w->rank == 2;
        const int cout = linear ? (int)w->ne[1] : (int)w->ne[2];
        if (linear ? ((int)w->ne[0] != cin || K != 1) : ((int)w->ne[1] != cin || (int)w->ne[0] != K)) {
            err = "exact stage one: unexpected shape of " + wname;
            return -1;
        }
        prof.begin(label, 0, 0, stream);
        hipError_t e = launch_exact_conv(x, w->d, b->d, y, res, dl, B, cin, cout, K, 1, pad_l, Tmax, relu, scale, stream);
        prof.end(stream);
        if (e != hipSuccess) {
            err = std::string("launch_exact_conv: ") + hipGetErrorString(e);
            return -1;
        }
        return 0;
    };
    auto ln_op = [&](TensorRef x, const std::string& gname, const std::string& bname, float eps, bool gelu) -> int {
        const ExactTensor* g = W(gname);
        const ExactTensor* b = W(bname);
        if (!g || !b) return -1;
        prof.begin("exact_layer_norm", 0, 0, stream);
        hipError_t e = launch_exact_layer_norm(x, g->d, b->d, dl, B, H, Tmax, eps, gelu ? gelu_tab : nullptr, stream);
        prof.end(stream);
        if (e != hipSuccess) {
            err = std::string("launch_exact_layer_norm: ") + hipGetErrorString(e);
            return -1;
        }
        return 0;
    };

    // ---- text encoder (vits.cpp:244-440) ----
    c.rx.phase("vits.text_encoder");
    TensorRef x = TR(s1.x, H, ts), qkv = TR(s1.qkv, 3 * H, ts), att = TR(s1.att, H, ts), tmp = TR(s1.tmp, H, ts), ffn = TR(s1.ffn, hp.ffn_dim, ts);
    {
        const ExactTensor* emb = W("text_encoder.embed_tokens.weight");
        if (!emb) return -1;
        prof.begin("embed", 0, 0, stream);
        HIP_OK(launch_embed(s1.ids, id_stride, dl, const_cast<float*>(emb->d), H, (float)std::sqrt((double)H), x, B, Tmax, stream));  // :263 one multiply per element
        prof.end(stream);
    }
    const float scaling = (float)std::pow((double)hd, -0.5);  // :296
    for (int l = 0; l < hp.layers; ++l) {
        const std::string base = "text_encoder.encoder.layers." + std::to_string(l) + ".";
        TensorRef q = sub(qkv, 0), k = sub(qkv, H), v = sub(qkv, 2 * H);
        q.bs = k.bs = v.bs = qkv.bs;
        if (conv_op("exact_linear", x, H, base + "attention.q_proj.weight", base + "attention.q_proj.bias", q, 1, 0, false, &scaling, none)) return -1;
        if (conv_op("exact_linear", x, H, base + "attention.k_proj.weight", base + "attention.k_proj.bias", k, 1, 0, false, nullptr, none)) return -1;
        if (conv_op("exact_linear", x, H, base + "attention.v_proj.weight", base + "attention.v_proj.bias", v, 1, 0, false, nullptr, none)) return -1;
        const ExactTensor* ek = W(base + "attention.emb_rel_k");
        const ExactTensor* ev = W(base + "attention.emb_rel_v");
        if (!ek || !ev) return -1;
        // (q, k, v are row ranges of one [3H]-row buffer: its batch stride; att has H rows — the kernel takes separate batch strides, one row stride)
        TensorRef att3 = att;
        prof.begin("exact_attention", 0, 0, stream);
        hipError_t e = launch_exact_attention(q, k, v, ek->d, ev->d, att3, s1.ex_scores, ts, dl, B, heads, hd, Tmax, hp.window, exp_tab, stream);
        prof.end(stream);
        if (e != hipSuccess) {
            err = std::string("launch_exact_attention: ") + hipGetErrorString(e);
            return -1;
        }
        // x = x + out_proj(att) (:358, :367), LayerNorm (:365-372)
        if (conv_op("exact_linear", att, H, base + "attention.out_proj.weight", base + "attention.out_proj.bias", tmp, 1, 0, false, nullptr, x)) return -1;
        if (ln_op(tmp, base + "layer_norm.weight", base + "layer_norm.bias", hp.ln_eps, false)) return -1;
        // feed forward (:377-407): conv k, relu, conv k, + residual (:416), LayerNorm (:412-418); tmp holds the layer's normalised input
        const int pl = (hp.ffn_k - 1) / 2;
        if (conv_op("exact_conv", tmp, H, base + "feed_forward.conv_1.weight", base + "feed_forward.conv_1.bias", ffn, hp.ffn_k, pl, true, nullptr, none)) return -1;
        if (conv_op("exact_conv", ffn, hp.ffn_dim, base + "feed_forward.conv_2.weight", base + "feed_forward.conv_2.bias", x, hp.ffn_k, pl, false, nullptr, tmp)) return -1;
        if (ln_op(x, base + "final_layer_norm.weight", base + "final_layer_norm.bias", hp.ln_eps, false)) return -1;
    }
    TensorRef stats = TR(s1.stats, 2 * F, ts);
    if (conv_op("exact_conv", x, H, "text_encoder.project.weight", "text_encoder.project.bias", stats, 1, 0, false, nullptr, none)) return -1;  // :429
    if (o.collect_taps) {
        snapshot("enc_out", x, H, Tmax, B, tlen);
        snapshot("prior_mean", sub(stats, 0), F, Tmax, B, tlen);
        snapshot("prior_logvar", sub(stats, F), F, Tmax, B, tlen);
    }

    // ---- stochastic duration predictor, reverse (vits.cpp:927-972) ----
    c.rx.phase("vits.duration_predictor");
    TensorRef dpx = TR(s1.dpx, H, ts), dpy = TR(s1.dpy, H, ts), dpp = TR(s1.dpp, H, ts), cond = TR(s1.cond, H, ts), z = TR(s1.z, 2, ts), u = TR(s1.u, 32, ts);
    // DDS block (:646-692) on xx in place; hh, pp scratch
    auto dds_op = [&](const std::string& base, TensorRef xx, TensorRef hh, TensorRef pp) -> int {
        int dil = 1;
        for (int i = 0; i < hp.dds_layers; ++i) {
            const std::string si = std::to_string(i);
            const ExactTensor* wd = W(base + "convs_dilated." + si + ".weight");
            const ExactTensor* bd = W(base + "convs_dilated." + si + ".bias");
            if (!wd || !bd) return -1;
            const int pad = (hp.dp_k * dil - dil) / 2;  // :660
            prof.begin("exact_depthwise", 0, 0, stream);
            hipError_t e = launch_exact_depthwise(xx, wd->d, bd->d, hh, dl, B, H, hp.dp_k, dil, pad, Tmax, stream);
            prof.end(stream);
            if (e != hipSuccess) {
                err = std::string("launch_exact_depthwise: ") + hipGetErrorString(e);
                return -1;
            }
            if (ln_op(hh, base + "norms_1." + si + ".weight", base + "norms_1." + si + ".bias", 1e-5f, true)) return -1;  // :668-673
            if (conv_op("exact_conv", hh, H, base + "convs_pointwise." + si + ".weight", base + "convs_pointwise." + si + ".bias", pp, 1, 0, false, nullptr, none)) return -1;
            if (ln_op(pp, base + "norms_2." + si + ".weight", base + "norms_2." + si + ".bias", 1e-5f, true)) return -1;   // :679-687
            prof.begin("exact_add", 0, 0, stream);
            e = launch_exact_add(xx, pp, dl, B, H, Tmax, stream);  // :688
            prof.end(stream);
            if (e != hipSuccess) {
                err = std::string("launch_exact_add: ") + hipGetErrorString(e);
                return -1;
            }
            dil *= hp.dp_k;  // :659
        }
        return 0;
    };
    const std::string dp = "duration_predictor.";
    if (conv_op("exact_conv", x, H, dp + "conv_pre.weight", dp + "conv_pre.bias", dpx, 1, 0, false, nullptr, none)) return -1;  // :934
    if (dds_op(dp + "conv_dds.", dpx, dpy, dpp)) return -1;                                                                        // :941
    if (conv_op("exact_conv", dpx, H, dp + "conv_proj.weight", dp + "conv_proj.bias", cond, 1, 0, false, nullptr, none)) return -1;  // :943
    std::vector<float> host_noise;
    if (o.noise_kind == VITS_NOISE_COUNTER) {
        prof.begin("noise_dur", 0, 0, stream);
        HIP_OK(launch_noise_dur(z, dl, B, Tmax, o.noise_seed, s1.seed_off, hp.noise_scale_dur, stream));  // one multiply per element (:948-949)
        prof.end(stream);
    } else if (c.ref_ahead) {
        // (batch 1, reference noise: the helper thread of engine.cpp holds the stream — take the [T, 2] tensor from it)
        host_noise.assign((size_t)2 * ts, 0.f);
        const float* tmpn = c.ref_ahead->duration_noise();
        for (int ch = 0; ch < 2; ++ch) std::memcpy(&host_noise[(size_t)ch * ts], tmpn + (size_t)ch * tlen[0], sizeof(float) * tlen[0]);
        HIP_OK(hipMemcpyAsync(s1.z, host_noise.data(), sizeof(float) * host_noise.size(), hipMemcpyHostToDevice, stream));
        prof.fence();
        if (o.collect_taps) snapshot("noise_dur", z, 2, Tmax, B, tlen);
        HIP_OK(launch_scale_rows(z, 2, hp.noise_scale_dur, B, Tmax, stream));
        HIP_OK(hipStreamSynchronize(stream));  // host_noise goes out of scope
    } else {
        host_noise.assign((size_t)B * 2 * ts, 0.f);
        for (int b = 0; b < B; ++b) {
            if (o.noise_kind == VITS_NOISE_EXPLICIT) {
                if (!o.noise_dur) {
                    err = "noise_dur missing";
                    return -1;
                }
                for (int ch = 0; ch < 2; ++ch) std::memcpy(&host_noise[((size_t)b * 2 + ch) * ts], o.noise_dur + ((size_t)b * 2 + ch) * id_stride, sizeof(float) * tlen[b]);
            } else {
                std::vector<float> tmpn((size_t)2 * tlen[b]);
                reference_noise_fill(tmpn.data(), tmpn.size());
                for (int ch = 0; ch < 2; ++ch) std::memcpy(&host_noise[((size_t)b * 2 + ch) * ts], &tmpn[(size_t)ch * tlen[b]], sizeof(float) * tlen[b]);
            }
        }
        HIP_OK(hipMemcpyAsync(s1.z, host_noise.data(), sizeof(float) * host_noise.size(), hipMemcpyHostToDevice, stream));
        prof.fence();
        if (o.collect_taps) snapshot("noise_dur", z, 2, Tmax, B, tlen);
        HIP_OK(launch_scale_rows(z, 2, hp.noise_scale_dur, B, Tmax, stream));
        HIP_OK(hipStreamSynchronize(stream));  // host_noise goes out of scope
    }
    const int nb = hp.dp_bins;
    const float inv_sqrt = (float)(1.0 / std::sqrt((double)H));                   // :877
    const float constant = (float)std::log(std::exp(1.0 - (double)1e-3f) - 1.0);  // :826 (host, double, as in the oracle)
    int c_first = 0;  // physical row of z holding logical latent channel 0 (the flips of :956 are index swaps)
    for (int fl = hp.dp_flows; fl > -1; --fl) {
        if (fl == 1) continue;
        c_first ^= 1;
        const std::string fb = dp + "flows." + std::to_string(fl) + ".";
        if (fl == 0) {
            // elementwise affine, reverse (:901-925; Q5): e = exp(+-log_scale) is two numbers per model, computed on the HOST on both sides
            std::vector<float> tr, ls;
            for (const TensorEntry& t : exact_src_) {
                if (t.name == fb + "translate") tr = t.to_f32();
                if (t.name == fb + "log_scale") ls = t.to_f32();
            }
            if (tr.size() < 2 || ls.size() < 2) {
                err = "[ERROR] tensor not found: " + fb + "translate / log_scale";
                return -1;
            }
            const float e0 = std::exp(refmode ? ls[0] : -ls[0]), e1 = std::exp(refmode ? ls[1] : -ls[1]);
            prof.begin("exact_affine", 0, 0, stream);
            HIP_OK(launch_exact_affine(z, c_first, tr[0], tr[1], e0, e1, dl, B, Tmax, stream));
            prof.end(stream);
            continue;
        }
        // conv flow (:855-899): h = conv_pre(z0) (1 -> H), DDS(h + cond), u = conv_proj(h), spline step on z1
        TensorRef z0 = sub(z, c_first);
        z0.bs = z.bs;
        if (conv_op("exact_conv", z0, 1, fb + "conv_pre.weight", fb + "conv_pre.bias", dpy, 1, 0, false, nullptr, none)) return -1;  // :864
        prof.begin("exact_add", 0, 0, stream);
        HIP_OK(launch_exact_add(dpy, cond, dl, B, H, Tmax, stream));  // :651-653
        prof.end(stream);
        if (dds_op(fb + "conv_dds.", dpy, dpx, dpp)) return -1;  // :868
        {
            const ExactTensor* w = W(fb + "conv_proj.weight");
            if (!w || (int)w->ne[2] != 3 * nb - 1 || 3 * nb - 1 > 32) {
                if (w) err = "exact stage one: unexpected shape of " + fb + "conv_proj.weight";
                return -1;
            }
        }
        if (conv_op("exact_conv", dpy, H, fb + "conv_proj.weight", fb + "conv_proj.bias", u, 1, 0, false, nullptr, none)) return -1;  // :871
        prof.begin("exact_spline", 0, 0, stream);
        HIP_OK(launch_exact_spline(z, 1 - c_first, u, s1.ex_tok, s1.ex_tok + (size_t)B * ts, s1.ex_tok + (size_t)2 * B * ts, ts, dl, B, Tmax, nb, hp.dp_tail, inv_sqrt, constant,
                                   refmode, exp_tab, stream));
        prof.end(stream);
    }
    if (o.collect_taps) snapshot("log_duration", sub(z, c_first), 1, Tmax, B, tlen);
    prof.begin("durations", 0, 0, stream);
    HIP_OK(launch_durations(z, c_first, dl, B, id_stride, (float)(1.0 / hp.speaking_rate), o.fixed_duration, s1.dur, s1.cum, s1.frames, s1.stage_lens, n_up + 1, s1.stage_mul,
                            s1.stage_add, stream, /*exact=*/true));
    prof.end(stream);
    c.c_first = c_first;
    return 0;
}

}  // namespace vits
