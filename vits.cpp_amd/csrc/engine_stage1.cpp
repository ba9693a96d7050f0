// engine_stage1.cpp — stage one of a call: text encoder (vits.cpp:244-440) and stochastic duration predictor, reverse
// (vits.cpp:927-972), up to the durations kernel. Sized by B x T; everything here is queued before the call's only host read.
#include "engine_internal.h"

namespace vits {

// DDS block (vits.cpp:646-692): x is updated in place; y, p are scratch [B][H][ts]
hipError_t Engine::run_dds(const DdsW& d, TensorRef x, TensorRef y, TensorRef p, const int* lens, int batch, int tmax, int64_t sum_t) {
    const int H = hp.hidden;
    TensorRef none;
    int dil = 1;
    // Each layer as ONE kernel (misc_kernels.hip dds_layer_kernel; bit-identical to the three launches below). A fused block reads a
    // halo of its neighbours' columns, so a layer never writes the buffer it reads: x -> y -> p -> ... -> x.
    const int n = hp.dds_layers;
    bool fuse = n >= 2 && !knobs.no_dds_fuse;
    for (int i = 0, dl = 1; i < n && fuse; ++i, dl *= hp.dp_k) fuse = dds_layer_supported(d.pw[i], H, hp.dp_k, dl, arith_now_);
    if (fuse) {
        TensorRef src = x;
        for (int i = 0; i < n; ++i) {
            TensorRef dst = i == n - 1 ? x : (src.p == y.p ? p : y);
            prof.begin("dds_layer_fused", 2.0 * H * H * (double)sum_t, 8.0 * H * (double)sum_t + (double)(arith_now_ == VITS_ARITH_F32 ? d.pw[i].bytes : d.pw[i].bytes16), stream);
            hipError_t e = launch_dds_layer(src, dst, d.dw_w[i], d.dw_b[i], d.n1_g[i], d.n1_b[i], d.pw[i], d.n2_g[i], d.n2_b[i], lens, batch, H, tmax, hp.dp_k, dil, 1e-5f,
                                            arith_now_, stream, ggml_tabs_);
            prof.end(stream);
            if (e != hipSuccess) return e;
            src = dst;
            dil *= hp.dp_k;
        }
        return hipSuccess;
    }
    for (int i = 0; i < hp.dds_layers; ++i) {
        KPROF("dds_depthwise_ln_gelu", launch_dds_depthwise(x, none, d.dw_w[i], d.dw_b[i], d.n1_g[i], d.n1_b[i], y, lens, batch, H, tmax, hp.dp_k, dil, 1e-5f, stream, arith_now_, ggml_tabs_));
        ConvCall c;
        c.x = y;
        c.y = p;
        c.len_in = lens;
        c.len_out = lens;
        c.batch = batch;
        c.t_in = c.t_out = tmax;
        c.sum_in = c.sum_out = sum_t;
        hipError_t e = conv("conv1x1_dp", d.pw[i], c);
        if (e != hipSuccess) return e;
        KPROF("ln_gelu_residual", launch_add_layer_norm(p, none, d.n2_g[i], d.n2_b[i], none, lens, batch, H, tmax, 1e-5f, 1, x, stream, ggml_tabs_));
        dil *= hp.dp_k;
    }
    return hipSuccess;
}

// Small grids (batch 1, a few short utterances): every DDS layer on the 16-token latency kernel, the per-token ops in front of and behind the block
// inside its first / last layer (stage1_lat.hip). Same floats as run_dds + the separate launches (tests/test_gpu_edge_and_scale.py).
bool Engine::dds_lat_ok(const DdsW& d, const DdsEnds& e, int batch, int tmax) const {
    if (knobs.kernel.no_dds_lat || knobs.no_dds_fuse || arith_now_ != VITS_ARITH_F32 || hp.dds_layers < 2) return false;
    if ((int64_t)batch * ((tmax + 15) / 16) > knobs.kernel.dds_lat_max_blocks) return false;
    for (int i = 0, dl = 1; i < hp.dds_layers; ++i, dl *= hp.dp_k)
        if (!dds_layer_lat_supported(d.pw[i], hp.hidden, hp.dp_k, dl)) return false;
    auto conv_ok = [&](const PackedConv* c, bool square) {
        return !c || (c->cin == hp.hidden && c->kt == 1 && c->epi == EPI_STD && c->bias && c->wp_l16 && (square ? c->cout == hp.hidden : c->cout <= hp.hidden));
    };
    return conv_ok(e.head_conv, true) && conv_ok(e.tail_conv, false);
}

hipError_t Engine::run_dds_lat(const DdsW& d, const DdsEnds& e, TensorRef a, TensorRef b, const int* lens, int batch, int tmax, int64_t sum_t) {
    const int H = hp.hidden, n = hp.dds_layers;
    TensorRef src;
    int dil = 1;
    for (int i = 0; i < n; ++i) {
        DdsLatCall c;
        c.dw_w = d.dw_w[i], c.dw_b = d.dw_b[i], c.g1 = d.n1_g[i], c.b1 = d.n1_b[i], c.g2 = d.n2_g[i], c.b2 = d.n2_b[i];
        c.pw = &d.pw[i];
        c.lens = lens;
        c.batch = batch, c.channels = H, c.tmax = tmax, c.k = hp.dp_k, c.dil = dil;
        c.tabs = ggml_tabs_;
        double flop = 2.0 * H * H * (double)sum_t, bytes = 8.0 * H * (double)sum_t + (double)d.pw[i].bytes;
        if (i == 0) {
            if (e.head_w) {
                c.head_w = e.head_w, c.head_b = e.head_b, c.z = e.z, c.cond = e.cond, c.zc = e.zc;
            } else if (e.head_conv) {
                c.head_conv = e.head_conv, c.x = e.head_x;
                flop += 2.0 * H * H * (double)sum_t;
                bytes += (double)e.head_conv->bytes;
            } else
                c.x = e.head_x;
        } else
            c.x = src;
        TensorRef dst = src.p == a.p ? b : a;
        if (i == n - 1 && e.tail_conv) {
            c.tail_conv = e.tail_conv, c.y2 = e.tail_y;
            flop += 2.0 * e.tail_conv->cout * H * (double)sum_t;
            bytes += (double)e.tail_conv->bytes;
        } else
            c.y = i == n - 1 ? e.tail_y : dst;
        prof.begin("dds_layer_lat", flop, bytes, stream);
        hipError_t err = launch_dds_layer_lat(c, stream);
        prof.end(stream);
        if (err != hipSuccess) return err;
        src = dst;
        dil *= hp.dp_k;
    }
    return hipSuccess;
}

int Engine::layout_stage_one(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, id_stride = c.id_stride, ts = c.ts, n_up = c.n_up;
    const int H = hp.hidden, F = hp.flow_size;
    Call::S1& s1 = c.s1;
    size_t x16_elems1 = 0;
    const size_t hdr_ints = (size_t)B * id_stride + 2 * (size_t)B + 2 * (size_t)(n_up + 1);
    auto layout1 = [&](Arena& a) {
        // host-written header, one block = one H2D copy: ids | lens | stage_mul | stage_add | seed_off
        s1.ids = a.alloc<int>(hdr_ints);
        s1.lens = s1.ids + (size_t)B * id_stride;
        s1.stage_mul = s1.lens + B;
        s1.stage_add = s1.stage_mul + (n_up + 1);
        s1.seed_off = s1.stage_add + (n_up + 1);
        s1.cum = a.alloc<int>((size_t)B * id_stride);
        s1.frames = a.alloc<int>(B);
        s1.stage_lens = a.alloc<int>((size_t)(n_up + 1) * B);
        s1.dur = a.alloc<float>((size_t)B * id_stride);
        s1.x = a.alloc<float>((size_t)B * H * ts);
        s1.qkv = a.alloc<float>((size_t)B * 3 * H * ts);
        s1.att = a.alloc<float>((size_t)B * H * ts);
        s1.tmp = a.alloc<float>((size_t)B * H * ts);
        s1.ffn = a.alloc<float>((size_t)B * hp.ffn_dim * ts);
        s1.stats = a.alloc<float>((size_t)B * 2 * F * ts);
        s1.dpx = a.alloc<float>((size_t)B * H * ts);
        s1.dpy = a.alloc<float>((size_t)B * H * ts);
        s1.dpp = a.alloc<float>((size_t)B * H * ts);
        s1.cond = a.alloc<float>((size_t)B * H * ts);
        s1.z = a.alloc<float>((size_t)B * 2 * ts);
        s1.u = a.alloc<float>((size_t)B * 32 * ts);
        if (ggml_tables == 1) {  // exact-order stage one: a score row per (utterance, head, query), three token rows per utterance for the spline step
            s1.ex_scores = a.alloc<float>((size_t)B * hp.heads * c.Tmax * ts);
            s1.ex_tok = a.alloc<float>((size_t)3 * B * ts);
        }
        // 16-bit arithmetic modes: scratch for the rounded copy of a conv input (largest c_in of stage one)
        x16_elems1 = arith_now_ != VITS_ARITH_F32 ? (size_t)B * round_up(std::max({hp.ffn_dim, 2 * F, H}), 8) * round_up(ts, 8) : 0;
        s1.x16 = x16_elems1 ? a.alloc<uint16_t>(x16_elems1) : nullptr;
    };
    {
        Arena measure;
        measure.cap = (size_t)1 << 60;
        layout1(measure);
        const size_t need = measure.off + 4096;
        measure.cap = 0;
        if (need > a1().cap) HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(a1().reserve(need));
        layout1(a1());
        for (int i = 0; i < 3; ++i) {
            x16_[i] = Ref16();
            x16_cap_[i] = 0;
        }
        x16_[0].p = s1.x16;
        x16_cap_[0] = x16_elems1;
    }
    // vocoder stage lengths as affine functions of the frame count L: len_i = L*mul_i + add_i (Q1: the reference
    // never crops the transposed conv, so every stage gains K - s samples; vits.cpp:187)
    std::vector<int>& smul = c.smul;
    std::vector<int>& sadd = c.sadd;
    smul.assign(n_up + 1, 0);
    sadd.assign(n_up + 1, 0);
    smul[0] = 1;
    sadd[0] = 0;
    for (int i = 0; i < n_up; ++i) {
        const int s = ups_[i].stride, K = ups_[i].k;
        const int crop = c.refmode ? 0 : (K - s) / 2;
        smul[i + 1] = smul[i] * s;
        sadd[i + 1] = sadd[i] * s + (K - s - 2 * crop);
    }
    const int32_t* ids = c.ids;
    const std::vector<int>& tlen = c.tlen;
    {
        // The header travels through engine-owned PINNED memory (two slots, each guarded by an event): an async call may
        // return while the copy is still queued, so neither the caller's ids nor locals of this function may be its source.
        HStage& hs = hstage_[hstage_next_];
        hstage_next_ ^= 1;
        if (hs.pending) HIP_OK(hipEventSynchronize(hs.ev));
        hs.pending = false;
        if (!hs.ev) HIP_OK(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
        if (hs.cap < hdr_ints) {
            if (hs.p) hipHostFree(hs.p);
            hs.p = nullptr;
            hs.cap = 0;
            HIP_OK(hipHostMalloc((void**)&hs.p, (hdr_ints + hdr_ints / 4 + 64) * sizeof(int), hipHostMallocDefault));
            hs.cap = hdr_ints + hdr_ints / 4 + 64;
        }
        std::memcpy(hs.p, ids, sizeof(int) * (size_t)B * id_stride);
        std::memcpy(hs.p + (size_t)B * id_stride, tlen.data(), sizeof(int) * B);
        std::memcpy(hs.p + (size_t)B * id_stride + B, smul.data(), sizeof(int) * (n_up + 1));
        std::memcpy(hs.p + (size_t)B * id_stride + B + (n_up + 1), sadd.data(), sizeof(int) * (n_up + 1));
        // counter-noise stream of utterance b: noise_seed + seed_off[b] (default b; a dispatcher that re-orders utterances
        // across ranks passes each one's global index so that its audio does not depend on where it ran)
        for (int b = 0; b < B; ++b) hs.p[(size_t)B * id_stride + B + 2 * (n_up + 1) + b] = o.noise_seed_offsets ? o.noise_seed_offsets[b] : b;
        HIP_OK(hipMemcpyAsync(s1.ids, hs.p, sizeof(int) * hdr_ints, hipMemcpyHostToDevice, stream));
        HIP_OK(hipEventRecord(hs.ev, stream));
        hs.pending = true;
        prof.fence();
    }
    return 0;
}

// ---- text encoder (vits.cpp:244-440) ---------------------------------------------------------------------
int Engine::run_text_encoder(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, id_stride = c.id_stride, ts = c.ts, Tmax = c.Tmax;
    const int H = hp.hidden, F = hp.flow_size, heads = hp.heads, hd = H / heads;
    const int64_t sum_t = c.sum_t;
    const std::vector<int>& tlen = c.tlen;
    Call::S1& s1 = c.s1;
    const int* dl = s1.lens;
    TensorRef none;
    auto TR = make_ref;
    auto sub = sub_rows;
    TensorRef x = TR(s1.x, H, ts), qkv = TR(s1.qkv, 3 * H, ts), att = TR(s1.att, H, ts), tmp = TR(s1.tmp, H, ts), ffn = TR(s1.ffn, hp.ffn_dim, ts);
    auto mk = [&](TensorRef xin, TensorRef yout, int tmax_) {
        ConvCall c;
        c.x = xin;
        c.y = yout;
        c.len_in = dl;
        c.len_out = dl;
        c.batch = B;
        c.t_in = c.t_out = tmax_;
        c.sum_in = c.sum_out = sum_t;
        return c;
    };

    c.rx.phase("vits.text_encoder");
    prof.begin("embed", 0, 0, stream);
    HIP_OK(launch_embed(s1.ids, id_stride, dl, emb_, H, (float)std::sqrt((double)H), x, B, Tmax, stream));
    prof.end(stream);
    const float q_scale = (float)std::pow((double)hd, -0.5);
    // A LayerNorm whose only consumer is a conv that will run on conv_lat16_kernel (tiny grids: batch 1, a few short utterances) is applied by that conv
    // ON LOAD (ConvCall::ln_gamma: same order of operations, the normalised tensor written to x by the conv's first row group) instead of being a launch of
    // its own: 12 of the text encoder's 43 launches at batch 1. `pending`: tmp holds a sum that still awaits its norm.
    struct PendingLn {
        const float *g = nullptr, *b = nullptr;
    } pending;
    auto norm_now = [&]() -> hipError_t {  // the separate launch: tmp -> x
        prof.begin("layer_norm", 0, 0, stream);
        hipError_t e = launch_add_layer_norm(tmp, none, pending.g, pending.b, x, dl, B, H, Tmax, hp.ln_eps, 0, none, stream);
        prof.end(stream);
        pending = PendingLn();
        return e;
    };
    // conv of LN(tmp) (or of x when nothing is pending); leaves the normalised tensor in x either way
    auto conv_of_norm = [&](const char* label, const PackedConv& w, ConvCall cc) -> hipError_t {
        if (pending.g) {
            ConvCall fused = cc;
            fused.x = tmp;
            fused.ln_gamma = pending.g, fused.ln_beta = pending.b, fused.ln_eps = hp.ln_eps, fused.ln_out = x;
            if (!knobs.kernel.no_ln_fuse && arith_now_ == VITS_ARITH_F32 && conv_ln_on_load_ok(w, fused)) {
                pending = PendingLn();
                return conv(label, w, fused);
            }
            if (hipError_t e = norm_now()) return e;
        }
        cc.x = x;
        return conv(label, w, cc);
    };
    for (int l = 0; l < hp.layers; ++l) {
        const EncoderLayerW& L = enc_[l];
        HIP_OK(conv_of_norm("enc_qkv_gemm", L.qkv, mk(x, qkv, Tmax)));
        prof.begin("rel_attention", 0, 0, stream);
        HIP_OK(launch_rel_attention(sub(qkv, 0), sub(qkv, H), sub(qkv, 2 * H), L.rel_k, L.rel_v, att, dl, B, heads, hd, Tmax, hp.window, q_scale, stream, ggml_tabs_));
        prof.end(stream);
        {
            ConvCall c = mk(att, tmp, Tmax);
            c.res = x;  // residual + attention output (vits.cpp:367)
            HIP_OK(conv("enc_out_gemm", L.out, c));
        }
        pending.g = L.ln1_g, pending.b = L.ln1_b;  // :365-372
        {
            ConvCall c = mk(x, ffn, Tmax);
            c.pad_l = (hp.ffn_k - 1) / 2;  // vits.cpp:388
            c.post_act = 1;                // relu :397
            HIP_OK(conv_of_norm("enc_ffn_conv", L.ffn1, c));
        }
        {
            ConvCall c = mk(ffn, tmp, Tmax);
            c.pad_l = (hp.ffn_k - 1) / 2;
            c.res = x;  // :416
            HIP_OK(conv("enc_ffn_conv", L.ffn2, c));
        }
        pending.g = L.ln2_g, pending.b = L.ln2_b;  // :412-418
    }
    TensorRef stats = TR(s1.stats, 2 * F, ts);
    HIP_OK(conv_of_norm("enc_project", enc_proj_, mk(x, stats, Tmax)));  // :429 ; split :436 = channel ranges [0,F) and [F,2F)
    if (o.collect_taps) {
        snapshot("enc_out", x, H, Tmax, B, tlen);
        snapshot("prior_mean", sub(stats, 0), F, Tmax, B, tlen);
        snapshot("prior_logvar", sub(stats, F), F, Tmax, B, tlen);
    }
    return 0;
}

// ---- stochastic duration predictor, reverse (vits.cpp:927-972) ----------------------------------------
int Engine::run_duration_predictor(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, id_stride = c.id_stride, ts = c.ts, Tmax = c.Tmax, n_up = c.n_up, md = c.md;
    const bool refmode = c.refmode;
    const int H = hp.hidden;
    const int64_t sum_t = c.sum_t;
    const std::vector<int>& tlen = c.tlen;
    Call::S1& s1 = c.s1;
    const int* dl = s1.lens;
    auto TR = make_ref;
    auto sub = sub_rows;
    TensorRef x = TR(s1.x, H, ts);
    auto mk = [&](TensorRef xin, TensorRef yout, int tmax_) {
        ConvCall c;
        c.x = xin;
        c.y = yout;
        c.len_in = dl;
        c.len_out = dl;
        c.batch = B;
        c.t_in = c.t_out = tmax_;
        c.sum_in = c.sum_out = sum_t;
        return c;
    };
    c.rx.phase("vits.duration_predictor");
    TensorRef dpx = TR(s1.dpx, H, ts), dpy = TR(s1.dpy, H, ts), dpp = TR(s1.dpp, H, ts), cond = TR(s1.cond, H, ts), z = TR(s1.z, 2, ts), u = TR(s1.u, 32, ts);
    {
        DdsEnds e;  // conv_pre (vits.cpp:939) -> DDS block -> conv_proj (:941): three launches on small grids
        e.head_conv = &dp_pre_, e.head_x = x, e.tail_conv = &dp_proj_, e.tail_y = cond;
        if (dds_lat_ok(dp_dds_, e, B, Tmax)) {
            HIP_OK(run_dds_lat(dp_dds_, e, dpy, dpp, dl, B, Tmax, sum_t));
        } else {
            HIP_OK(conv("conv1x1_dp", dp_pre_, mk(x, dpx, Tmax)));
            HIP_OK(run_dds(dp_dds_, dpx, dpy, dpp, dl, B, Tmax, sum_t));
            HIP_OK(conv("conv1x1_dp", dp_proj_, mk(dpx, cond, Tmax)));
        }
    }
    std::vector<float> host_noise;
    if (o.noise_kind == VITS_NOISE_COUNTER) {
        prof.begin("noise_dur", 0, 0, stream);
        HIP_OK(launch_noise_dur(z, dl, B, Tmax, o.noise_seed, s1.seed_off, hp.noise_scale_dur, stream));
        prof.end(stream);
    } else if (c.ref_ahead) {
        // the reference's own call: the [T, 2] draws (vits.cpp:948) come from the helper thread (it holds the stream's lock: nobody else may draw now),
        // through a small pinned block — a copy from pageable memory in the middle of stage one would make the host wait for the text encoder
        const size_t n = (size_t)2 * ts;
        if (dur_noise_cap_ < n) {
            if (dur_noise_pinned_) hipHostFree(dur_noise_pinned_);
            dur_noise_pinned_ = nullptr;
            dur_noise_cap_ = 0;
            HIP_OK(hipHostMalloc((void**)&dur_noise_pinned_, (n + 256) * sizeof(float), hipHostMallocDefault));
            dur_noise_cap_ = n + 256;
        }
        if (dur_noise_ev_) HIP_OK(hipEventSynchronize(dur_noise_ev_));  // (the previous call's copy out of this block)
        else HIP_OK(hipEventCreateWithFlags(&dur_noise_ev_, hipEventDisableTiming));
        const float* tmpn = c.ref_ahead->duration_noise();  // memory order [2][T]
        std::memset(dur_noise_pinned_, 0, sizeof(float) * n);
        for (int ch = 0; ch < 2; ++ch) std::memcpy(dur_noise_pinned_ + (size_t)ch * ts, tmpn + (size_t)ch * tlen[0], sizeof(float) * tlen[0]);
        HIP_OK(hipMemcpyAsync(s1.z, dur_noise_pinned_, sizeof(float) * n, hipMemcpyHostToDevice, stream));
        HIP_OK(hipEventRecord(dur_noise_ev_, stream));
        prof.fence();
        if (o.collect_taps) snapshot("noise_dur", z, 2, Tmax, B, tlen);
        HIP_OK(launch_scale_rows(z, 2, hp.noise_scale_dur, B, Tmax, stream));
    } else {
        host_noise.assign((size_t)B * 2 * ts, 0.f);
        for (int b = 0; b < B; ++b) {
            if (o.noise_kind == VITS_NOISE_EXPLICIT) {
                if (!o.noise_dur) {
                    err = "noise_dur missing";
                    return -1;
                }
                for (int c = 0; c < 2; ++c) std::memcpy(&host_noise[((size_t)b * 2 + c) * ts], o.noise_dur + ((size_t)b * 2 + c) * id_stride, sizeof(float) * tlen[b]);
            } else {
                std::vector<float> tmpn((size_t)2 * tlen[b]);  // tensor_randn{T,2,1}: memory order [2][T] (vits.cpp:948)
                reference_noise_fill(tmpn.data(), tmpn.size());
                for (int c = 0; c < 2; ++c) std::memcpy(&host_noise[((size_t)b * 2 + c) * ts], &tmpn[(size_t)c * tlen[b]], sizeof(float) * tlen[b]);
            }
        }
        HIP_OK(hipMemcpyAsync(s1.z, host_noise.data(), sizeof(float) * host_noise.size(), hipMemcpyHostToDevice, stream));
        prof.fence();
        if (o.collect_taps) snapshot("noise_dur", z, 2, Tmax, B, tlen);
        HIP_OK(launch_scale_rows(z, 2, hp.noise_scale_dur, B, Tmax, stream));
    }
    int c_first = 0;  // physical row holding logical latent channel 0
    const float inv_sqrt = (float)(1.0 / std::sqrt((double)H));
    for (int fl = hp.dp_flows; fl > -1; --fl) {
        if (fl == 1) continue;
        c_first ^= 1;  // flip (vits.cpp:956) is an index swap
        if (fl == 0) {
            prof.begin("dp_affine", 0, 0, stream);
            HIP_OK(launch_affine(z, c_first, dp_translate_, dp_logscale_, refmode ? +1 : -1, dl, B, Tmax, stream));  // Q5
            prof.end(stream);
        } else {
            const DpFlowW& W = dp_flows_[fl - 1];
            DdsEnds e;  // small grids: conv_pre + conditioning inside the first DDS layer's kernel, the projection inside the last one's
            e.head_w = W.pre_w, e.head_b = W.pre_b, e.z = z, e.cond = cond, e.zc = c_first, e.tail_conv = &W.proj, e.tail_y = u;
            if (dds_lat_ok(W.dds, e, B, Tmax)) {
                HIP_OK(run_dds_lat(W.dds, e, dpx, dpp, dl, B, Tmax, sum_t));
            } else {
                // conv_pre (1 -> H, vits.cpp:864) fused with "inputs + global_conditioning" of the DDS block (:651-653)
                prof.begin("dp_flow_pre", 0, 0, stream);
                HIP_OK(launch_pointwise_from1(z, c_first, W.pre_w, W.pre_b, cond, dpy, dl, B, H, Tmax, stream, arith_now_));
                prof.end(stream);
                HIP_OK(run_dds(W.dds, dpy, dpx, dpp, dl, B, Tmax, sum_t));
                HIP_OK(conv("conv1x1_dp", W.proj, mk(dpy, u, Tmax)));
            }
            prof.begin("dp_spline", 0, 0, stream);
            HIP_OK(launch_spline(u, z, 1 - c_first, dl, B, Tmax, hp.dp_bins, hp.dp_tail, inv_sqrt, md, stream, ggml_tabs_));
            prof.end(stream);
        }
    }
    if (o.collect_taps) snapshot("log_duration", sub(z, c_first), 1, Tmax, B, tlen);
    prof.begin("durations", 0, 0, stream);
    HIP_OK(launch_durations(z, c_first, dl, B, id_stride, (float)(1.0 / hp.speaking_rate), o.fixed_duration, s1.dur, s1.cum, s1.frames, s1.stage_lens, n_up + 1,
                            s1.stage_mul, s1.stage_add, stream));
    prof.end(stream);
    c.c_first = c_first;
    return 0;
}

}  // namespace vits
