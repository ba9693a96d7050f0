// engine_flow.cpp — stage two of a call up to the vocoder: arena layout (sized by B x L after the host read of the frame
// counts), prior sampling through the alignment (vits.cpp:1028-1064) and the residual coupling flow, reverse
// (vits.cpp:519-538,500-517,452-498).
#include <thread>

#include "engine_internal.h"

namespace vits {

int Engine::layout_stage_two(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, n_up = c.n_up;
    const int H = hp.hidden, F = hp.flow_size;
    const std::vector<int>& frames = c.frames;
    const int Lmax = c.Lmax;
    Call::S1& s1 = c.s1;
    Call::S2& s2 = c.s2;
    auto TR = make_ref;
    auto sub = sub_rows;
    const std::vector<Call::Win>& wins = c.wins;
    const bool windowed = c.windowed;
    const int Lw_max = c.Lw_max;
    const std::vector<int>& smul = c.smul;
    const std::vector<int>& sadd = c.sadd;
    const std::vector<int>& smax = c.smax;
    // ---- stage two buffers -------------------------------------------------------------------------------------
    const int ls = c.ls = round_up(Lmax, 32), lws = c.lws = round_up(Lw_max, 32);
    size_t& big = c.big;
    big = 0;
    std::vector<int>& sts = c.sts;
    sts.assign(n_up + 1, 0);
    for (int i = 0; i <= n_up; ++i) sts[i] = round_up(Lw_max * smul[i] + sadd[i], 32);
    for (int i = 0; i < n_up; ++i) big = std::max(big, (size_t)B * ups_[i].channels * sts[i + 1]);
    const bool need_noise_buf = o.noise_kind != VITS_NOISE_COUNTER;
    // 16-bit arithmetic modes: the vocoder runs in the group layout of conv16.hip when its channel counts allow it (with or without
    // collect_taps: every tap is a tensor both layouts produce in fp32 [c][t]); scratch for the converter path's rounded inputs
    const bool fast16 = c.fast16 = arith_now_ != VITS_ARITH_F32 && vocoder_group_ok_ && !knobs.no_group16;
    c.fuse16 = fast16 && !knobs.no_fuse16;  // resblock conv pairs of the narrow stages as one kernel
    const size_t x16_elems2 = arith_now_ != VITS_ARITH_F32 ? std::max({big, (size_t)B * round_up(H, 8) * round_up(ls, 8), (size_t)B * round_up(hp.up_init, 8) * round_up(lws, 8), (size_t)B * round_up(F, 8) * round_up(ls, 8)}) + 64 : 0;
    const int S_stride = c.S_stride = round_up(smax[n_up], 32);
    // VITS_ARITH_F32_SPLIT: three bf16 planes per conv input of the wide stages' resblocks (conv_split.hip): the stage input once, t and the stream per
    // concurrently running resblock
    size_t sp_elems = 0;
    if (split_on())
        for (int i = 0; i < n_up; ++i)
            if (ups_[i].channels >= 128 && ups_[i].channels % 32 == 0) sp_elems = std::max(sp_elems, (size_t)3 * B * ups_[i].channels * sts[i + 1] + 64);
    auto layout2 = [&](Arena& a) {
        s2.zp = a.alloc<float>((size_t)B * F * ls);
        s2.noise = need_noise_buf ? a.alloc<float>((size_t)B * F * ls) : nullptr;
        s2.hout = a.alloc<float>((size_t)B * 2 * H * ls);
        s2.gate = a.alloc<float>((size_t)B * H * ls);
        s2.h0 = a.alloc<float>((size_t)B * hp.up_init * lws);
        s2.win_lens = windowed ? a.alloc<int>(wins.size() * (size_t)(n_up + 2) * B) : nullptr;
        s2.bu = a.alloc<float>(big);
        s2.bul = a.alloc<float>(big);
        for (int j = 0; j < 3; ++j) {
            // one (y, t) pair per concurrently running resblock
            const bool own = j == 0 || (knobs.rb_streams > 1 && (size_t)j < hp.rb_k.size());
            s2.by[j] = own ? a.alloc<float>(big) : s2.by[0];
            s2.bt[j] = own ? a.alloc<float>(big) : s2.bt[0];
            s2.byl[j] = own ? a.alloc<float>(big) : s2.byl[0];
        }
        s2.bs = a.alloc<float>(big);
        s2.bs16 = fast16 ? a.alloc<float>(big / 2 + 64) : nullptr;
        for (int j = 0; j < 3; ++j) s2.x16[j] = (x16_elems2 && (j == 0 || (knobs.rb_streams > 1 && !fast16))) ? a.alloc<uint16_t>(x16_elems2) : nullptr;
        s2.sp_u = sp_elems ? a.alloc<uint16_t>(sp_elems) : nullptr;
        for (int j = 0; j < 3; ++j) {
            const bool own = j == 0 || (knobs.rb_streams > 1 && (size_t)j < hp.rb_k.size());
            s2.sp_t[j] = !sp_elems ? nullptr : (own ? a.alloc<uint16_t>(sp_elems) : s2.sp_t[0]);
            s2.sp_y[j] = !sp_elems ? nullptr : (own ? a.alloc<uint16_t>(sp_elems) : s2.sp_y[0]);
        }
        s2.pre = o.collect_taps ? a.alloc<float>((size_t)B * S_stride) : nullptr;
        s2.wave = a.alloc<float>((size_t)B * S_stride);
    };
    {
        Arena measure;
        measure.cap = (size_t)1 << 60;
        layout2(measure);
        const size_t need = measure.off + 4096;
        measure.cap = 0;
        if (need > a2_.cap) HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(a2_.reserve(need));
        layout2(a2_);
        for (int j = 0; j < 3; ++j) {
            x16_[j] = Ref16();
            x16_[j].p = s2.x16[j];
            x16_cap_[j] = s2.x16[j] ? x16_elems2 : 0;
        }
    }
    for (int i = 0; i <= n_up && i < 8; ++i) c.d_len_full[i] = s1.stage_lens + (size_t)i * B;
    (void)sub;
    (void)TR;
    (void)frames;
    return 0;
}

// ---- prior sampling through the alignment (vits.cpp:1028-1064) -------------------------------------------
int Engine::run_prior_sampling(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, n_up = c.n_up;
    const int H = hp.hidden, F = hp.flow_size;
    const std::vector<int>& frames = c.frames;
    const int Lmax = c.Lmax;
    Call::S1& s1 = c.s1;
    Call::S2& s2 = c.s2;
    auto TR = make_ref;
    auto sub = sub_rows;
    const int id_stride = c.id_stride, ls = c.ls;
    const int* dl = s1.lens;
    const bool need_noise_buf = o.noise_kind != VITS_NOISE_COUNTER;
    TensorRef stats = TR(s1.stats, 2 * F, c.ts);
    c.rx.phase("vits.prior_sampling");
    TensorRef zp = TR(s2.zp, F, ls), noise = TR(s2.noise, F, ls);
    if (c.ref_ahead) {
        // batch 1, reference noise: the tensor was (mostly) drawn while stage one ran (engine.cpp); it is one dense [F][L] block in pinned memory
        const int L = frames[0];
        const size_t n = (size_t)F * (size_t)L;
        if (n > ref_noise_cap_) {  // (more frames per id than the block was sized for: a larger block once the helper rests at the old capacity; keep what was drawn)
            while (c.ref_ahead->drawn() < c.ref_ahead->capacity()) std::this_thread::yield();
            float* bigger = nullptr;
            HIP_OK(hipHostMalloc((void**)&bigger, n * sizeof(float), hipHostMallocDefault));
            std::memcpy(bigger, ref_noise_pinned_, sizeof(float) * c.ref_ahead->drawn());
            float* old = ref_noise_pinned_;
            ref_noise_pinned_ = bigger;
            ref_noise_cap_ = n;
            c.ref_ahead->rebase(ref_noise_pinned_, ref_noise_cap_);
            c.ref_ahead->finish(n);
            hipHostFree(old);
        } else
            c.ref_ahead->finish(n);
        c.ref_ahead = nullptr;
        noise.cs = L;
        noise.bs = (int64_t)F * L;
        HIP_OK(hipMemcpyAsync(s2.noise, ref_noise_pinned_, sizeof(float) * n, hipMemcpyHostToDevice, stream));
        prof.fence();
        if (o.async) HIP_OK(hipStreamSynchronize(stream));  // (the pinned block is reused by the next call)
        if (o.collect_taps) snapshot("noise_prior", noise, F, Lmax, B, frames);
    } else if (need_noise_buf) {
        std::vector<float> hn((size_t)B * F * ls, 0.f);
        for (int b = 0; b < B; ++b) {
            const int L = frames[b];
            if (o.noise_kind == VITS_NOISE_EXPLICIT) {
                if (!o.noise_prior) {
                    err = "noise_prior missing";
                    return -1;
                }
                for (int c = 0; c < F; ++c)
                    std::memcpy(&hn[((size_t)b * F + c) * ls], o.noise_prior + ((size_t)b * F + c) * o.noise_prior_stride, sizeof(float) * std::min<int64_t>(L, o.noise_prior_stride));
            } else {
                std::vector<float> tmpn((size_t)F * L);  // tensor_randn_like(prior_means ne=[L,F]) (vits.cpp:1059)
                reference_noise_fill(tmpn.data(), tmpn.size());
                for (int c = 0; c < F; ++c) std::memcpy(&hn[((size_t)b * F + c) * ls], &tmpn[(size_t)c * L], sizeof(float) * L);
            }
        }
        HIP_OK(hipMemcpyAsync(s2.noise, hn.data(), sizeof(float) * hn.size(), hipMemcpyHostToDevice, stream));
        prof.fence();
        HIP_OK(hipStreamSynchronize(stream));  // hn goes out of scope
        if (o.collect_taps) snapshot("noise_prior", noise, F, Lmax, B, frames);
    }
    prof.begin("prior_sample_gather", 0, 0, stream);
    HIP_OK(launch_zp(sub(stats, 0), sub(stats, F), s1.cum, id_stride, dl, s1.frames, noise, o.noise_kind == VITS_NOISE_COUNTER ? VITS_NOISE_COUNTER : VITS_NOISE_EXPLICIT,
                     o.noise_seed, s1.seed_off, hp.noise_scale, zp, B, F, Lmax, stream));
    prof.end(stream);
    if (o.collect_taps) snapshot("z_p", zp, F, Lmax, B, frames);
    (void)n_up;
    (void)H;
    return 0;
}

// ---- residual coupling flow, reverse (vits.cpp:519-538,500-517,452-498) ------------------------------------
int Engine::run_flow(Call& c) {
    std::string& err = c.err;
    const vits_process_opts& o = c.o;
    const int B = c.B, n_up = c.n_up;
    const int H = hp.hidden, F = hp.flow_size;
    const std::vector<int>& frames = c.frames;
    const int Lmax = c.Lmax;
    Call::S1& s1 = c.s1;
    Call::S2& s2 = c.s2;
    auto TR = make_ref;
    auto sub = sub_rows;
    const int ls = c.ls;
    const int64_t sum_frames = c.sum_frames;
    TensorRef zp = TR(s2.zp, F, ls);
    c.rx.phase("vits.flow");
    const int* ll = c.d_len_full[0];
    TensorRef hout = TR(s2.hout, 2 * H, ls), gate = TR(s2.gate, H, ls);
    TensorRef hh = hout;  // channels [0,H) = h, [H,2H) = skip accumulator "outputs" (vits.cpp:460)
    auto mk2 = [&](TensorRef xin, TensorRef yout) {
        ConvCall c;
        c.x = xin;
        c.y = yout;
        c.len_in = ll;
        c.len_out = ll;
        c.batch = B;
        c.t_in = c.t_out = Lmax;
        c.sum_in = c.sum_out = sum_frames;
        return c;
    };
    // 16-bit modes, fused coupling layers: a layer is ONE launch of sum_b ceil(frames_b / 48) blocks, one block per CU at a time (238 VGPRs x six
    // waves) — 320 blocks at batch 64 x 225 frames = a full round on the 256 CUs and a second one on 64 of them, four times over. Utterances never
    // interact, so the batch is dealt out over two independent chains of launches (main stream + a side stream): while one chain's layer drains its
    // tail the other's blocks take the free CUs (4 x 320 blocks in ~5 rounds instead of 8). Same kernel, same operands per utterance: same bits.
    int chains = 1, chain_b0[3] = {0, B, B};
    bool all_fused = arith_now_ != VITS_ARITH_F32 && !knobs.no_flow_fuse;
    for (int i = 0; i < hp.n_flows && all_fused; ++i)
        all_fused = (int)flow_[i].in_layers.size() == hp.wn_layers && (int)flow_[i].res_skip.size() == hp.wn_layers &&
                    flow_couple16_supported(H, F / 2, hp.wn_k, hp.wn_rate, hp.wn_layers, flow_[i].pre, flow_[i].in_layers.data(), flow_[i].res_skip.data(), flow_[i].post);
    if (all_fused && !prof.on && side_[0] && B >= 2 && knobs.flow_chains > 1) {
        int64_t blocks = 0;
        for (int b = 0; b < B; ++b) blocks += (frames[b] + 47) / 48;
        if (blocks > knobs.flow_chain_min_blocks && blocks > kernel_knobs().flow_narrow_max) {
            // equal shares of the blocks (by utterance, in order)
            int64_t run = 0;
            int cut = 0;
            while (cut < B - 1 && 2 * (run + (frames[cut] + 47) / 48) <= blocks) run += (frames[cut++] + 47) / 48;
            if (cut >= 1) {
                chains = 2;
                chain_b0[1] = cut;
            }
        }
    }
    if (chains > 1) {
        HIP_OK(hipEventRecord(ev_fork_, stream));
        HIP_OK(hipStreamWaitEvent(side_[0], ev_fork_, 0));
    }
    for (int i = hp.n_flows - 1; i > -1; --i) {
        const FlowLayerW& Lw = flow_[i];
        const bool flipped = ((hp.n_flows - i) % 2) == 1;
        TensorRef x0 = sub(zp, flipped ? F / 2 : 0), x1 = sub(zp, flipped ? 0 : F / 2);
        // 16-bit modes: the whole coupling layer as ONE kernel (flow_couple16_kernel, wavenet32.hip; bit-identical to the launches below)
        if (chains > 1) {
            for (int ch = 0; ch < chains; ++ch) {
                const int b0 = chain_b0[ch], nb = chain_b0[ch + 1] - b0;
                FlowCouple16Call fc;
                fc.x0 = x0;
                fc.x0.p += (int64_t)b0 * x0.bs;
                fc.x1 = x1;
                fc.x1.p += (int64_t)b0 * x1.bs;
                fc.lens = ll + b0;
                fc.batch = nb;
                fc.tmax = 0;
                for (int b = b0; b < b0 + nb; ++b) fc.tmax = std::max(fc.tmax, frames[b]);
                fc.hidden = H;
                fc.half = F / 2;
                HIP_OK(launch_flow_couple16(Lw.pre, Lw.in_layers.data(), Lw.res_skip.data(), Lw.post, fc, arith_now_, ch == 0 ? stream : side_[0]));
            }
            continue;
        }
        if (all_fused) {
            FlowCouple16Call fc;
            fc.x0 = x0;
            fc.x1 = x1;
            fc.lens = ll;
            fc.batch = B;
            fc.tmax = Lmax;
            fc.hidden = H;
            fc.half = F / 2;
            if (prof.on) {
                char full[160];
                std::snprintf(full, sizeof(full), "flow_coupling_layer|k%d|d1|C%d|e1|c%dx%d", hp.wn_k, H, F / 2, F / 2);
                double macs = (double)(F / 2) * H + (double)H * (F / 2), wbytes = (double)Lw.pre.bytes16 + (double)Lw.post.bytes16;
                for (int l = 0; l < hp.wn_layers; ++l) {
                    macs += (double)2 * H * H * hp.wn_k + (double)Lw.res_skip[l].cout * H;
                    wbytes += (double)Lw.in_layers[l].bytes16 + (double)Lw.res_skip[l].bytes16;
                }
                prof.begin(full, 2.0 * macs * (double)sum_frames, 4.0 * (double)sum_frames * (F / 2) * 3.0 + wbytes, stream, true);
            }
            HIP_OK(launch_flow_couple16(Lw.pre, Lw.in_layers.data(), Lw.res_skip.data(), Lw.post, fc, arith_now_, stream));
            prof.end(stream);
            continue;
        }
        HIP_OK(conv("flow_conv1x1", Lw.pre, mk2(x0, hh)));  // h -> hout[0,H)
        prof.begin("fill_zero", 0, 0, stream);
        HIP_OK(launch_fill_rows(sub(hout, H), H, 0.f, B, Lmax, stream));
        prof.end(stream);
        int dil = 1;
        // fp32: each WaveNet layer as ONE kernel (wavenet32.hip; bit-identical to the two launches below). A fused block reads a
        // 2-frame halo of its neighbours' h columns, so h alternates between hout[0,H) and the buffer the two-launch path uses for the
        // gate output; `outputs` (hout[H,2H)) is updated in place.
        // (large grids only: at batch 1 a layer is four blocks, and a block's six waves on four SIMDs run two MFMA chains deep:
        // 68 us against 39 us for the two launches with their split-gate tiles)
        bool fuse_wn = !knobs.no_wn_fuse && (ls & 3) == 0 && (int64_t)((Lmax + 31) / 32) * B >= 384 &&
                       (reinterpret_cast<uintptr_t>(hout.p) & 15) == 0 && (reinterpret_cast<uintptr_t>(gate.p) & 15) == 0;
        for (int l = 0; l < hp.wn_layers && fuse_wn; ++l) {
            int dl = 1;
            for (int q = 0; q < l; ++q) dl *= hp.wn_rate;
            fuse_wn = (arith_now_ == VITS_ARITH_F32 ? wavenet32_supported(H, hp.wn_k, dl, Lw.in_layers[l], Lw.res_skip[l])
                                               : wavenet16_supported(H, hp.wn_k, dl, Lw.in_layers[l], Lw.res_skip[l])) &&
                      Lw.res_skip[l].cout == (l + 1 < hp.wn_layers ? 2 * H : H);
        }
        if (fuse_wn) {
            TensorRef hcur = hh;  // rows [0,H) of hout
            for (int l = 0; l < hp.wn_layers; ++l) {
                WaveNet32Call w;
                w.h = hcur;
                w.h_out = hcur.p == gate.p ? hh : gate;
                if (l + 1 == hp.wn_layers) w.h_out = TensorRef();
                w.outputs = sub(hout, H);
                w.lens = ll;
                w.batch = B;
                w.tmax = Lmax;
                w.hidden = H;
                w.dil = 1;
                if (prof.on) {
                    char full[160];
                    std::snprintf(full, sizeof(full), "flow_wavenet_layer|k%d|d1|%c%d|e1|c%dx%d", hp.wn_k, arith_now_ == VITS_ARITH_F32 ? 'w' : 'W', H, H, Lw.res_skip[l].cout);
                    prof.begin(full, 2.0 * ((double)2 * H * H * hp.wn_k + (double)Lw.res_skip[l].cout * H) * (double)sum_frames,
                               4.0 * (double)sum_frames * (H + 2.0 * Lw.res_skip[l].cout) + (double)Lw.in_layers[l].bytes + (double)Lw.res_skip[l].bytes, stream, true);
                }
                if (arith_now_ == VITS_ARITH_F32) HIP_OK(launch_wavenet32(Lw.in_layers[l], Lw.res_skip[l], w, stream));
                else HIP_OK(launch_wavenet16(Lw.in_layers[l], Lw.res_skip[l], w, arith_now_, stream));
                prof.end(stream);
                if (w.h_out.p) hcur = w.h_out;
            }
        }
        for (int l = 0; l < hp.wn_layers && !fuse_wn; ++l) {
            ConvCall c = mk2(hh, gate);
            c.dil = dil;
            c.pad_l = (hp.wn_k * dil - dil) / 2;  // vits.cpp:470
            HIP_OK(conv("flow_wavenet_gated_conv", Lw.in_layers[l], c));
            if (l < hp.wn_layers - 1) {
                ConvCall r = mk2(gate, hout);  // rows [0,H): h += res ; rows [H,2H): outputs += skip (vits.cpp:484-489)
                r.res = hout;
                HIP_OK(conv("flow_conv1x1", Lw.res_skip[l], r));
            } else {
                ConvCall r = mk2(gate, sub(hout, H));  // outputs += res_skip (vits.cpp:491)
                r.res = sub(hout, H);
                HIP_OK(conv("flow_conv1x1", Lw.res_skip[l], r));
            }
            dil *= hp.wn_rate;
        }
        ConvCall pc = mk2(sub(hout, H), x1);  // x1 <- x1 - (W out + b): weights negated at load (vits.cpp:506,513)
        pc.res = x1;
        HIP_OK(conv("flow_conv1x1", Lw.post, pc));
    }
    if (chains > 1) {
        HIP_OK(hipEventRecord(ev_done_[0], side_[0]));
        HIP_OK(hipStreamWaitEvent(stream, ev_done_[0], 0));
    }
    if (o.collect_taps) snapshot("z_flow", zp, F, Lmax, B, frames);
    (void)n_up;
    (void)s1;
    return 0;
}

}  // namespace vits
