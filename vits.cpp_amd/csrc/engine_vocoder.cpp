// engine_vocoder.cpp — HiFiGAN (vits.cpp:583-644) over one window of frames: the 16-bit-operand path in the group layout of
// conv16.hip and the exact fp32 path. Window-local lengths throughout (a whole-utterance run is one window).
#include "engine_internal.h"

namespace vits {

int Engine::run_vocoder_window16(Call& c, WinCtx& w) {
    std::string& err = c.err;
    const int B = c.B, n_up = c.n_up;
    const int F = hp.flow_size;
    const bool refmode = c.refmode;
    Call::S2& s2 = c.s2;
    const std::vector<int>& sts = c.sts;
    const int lws = c.lws;
    const int Lw = w.Lw;
    const int* const* d_len = w.d_len;
    const std::vector<int>& smax = w.smax;
    const std::vector<int64_t>& ssum = w.ssum;
    const TensorRef zwin = w.zwin, pre = w.pre, wv = w.wv;
    const int emit_lo = w.emit_lo;
    const int* emit_hi = w.emit_hi;
    const float final_slope = w.final_slope;
    auto TR = make_ref;
    // ---- 16-bit-operand vocoder in the group layout of conv16.hip -----------------------------------------------
    // Every conv input is a 16-bit tensor WRITTEN by its producer (leaky_relu and rounding fused into the writer's
    // epilogue: what the reference's leaky_relu node + fp16 im2col compute, vits.cpp:554,567,613 + custom-ops.h:684-690);
    // the residual stream (vits.cpp:578) and the resblock sum (:622-635) stay fp32, in the same [c/8][t][8] layout.
    auto R16 = [](float* base, int channels, int stride) {
        Ref16 r;
        r.p = reinterpret_cast<uint16_t*>(base);
        r.ts = stride;
        r.bs = (int64_t)channels * stride;
        return r;
    };
    const size_t nk = hp.rb_k.size();
    Ref16 z16 = x16_[0];
    z16.ts = round_up(Lw, 8);
    z16.bs = (int64_t)(F / 8) * z16.ts * 8;
    prof.begin("to_group16", 0, 6.0 * F * (double)ssum[0], stream, true);
    HIP_OK(launch_to_group16(zwin, d_len[0], B, F, Lw, 1.0f, z16, arith_now_, stream));
    prof.end(stream);
    Ref16 cur16 = R16(s2.h0, hp.up_init, lws);
    {
        Conv16Call c;
        c.x = z16;
        c.len_in = c.len_out = d_len[0];
        c.batch = B;
        c.t_in = c.t_out = Lw;
        c.sum_in = c.sum_out = ssum[0];
        c.pad_l = (dec_pre_.kt - 1) / 2;
        c.y16 = cur16;
        c.y16_slope = hp.lrelu;  // only reader: the first upsampler, behind its leaky_relu (vits.cpp:613)
        HIP_OK(conv16("hifigan_conv_pre", dec_pre_, c, stream, 2.0 * (F + hp.up_init) * (double)ssum[0] + (double)dec_pre_.bytes16));
    }
    for (int i = 0; i < n_up; ++i) {
        const UpStageW& U = ups_[i];
        char rx_stage[32];
        std::snprintf(rx_stage, sizeof(rx_stage), "vits.hifigan.stage%d", i);
        RoctxRange rx_stage_range(rx_stage);
        const int C = U.channels, st_in = i, st_out = i + 1;
        const int64_t g_bs = (int64_t)C * sts[st_out];
        const int g_ts = sts[st_out];
        const double n_out = (double)C * (double)ssum[st_out];
        const Ref16 bul16 = R16(s2.bul, C, sts[st_out]), bsum16 = R16(s2.bs16, C, sts[st_out]);
        {
            Conv16Call c;
            c.x = cur16;
            c.len_in = d_len[st_in];
            c.len_out = d_len[st_out];
            c.batch = B;
            c.t_in = smax[st_in];
            c.t_out = smax[st_out];
            c.sum_in = ssum[st_in];
            c.sum_out = ssum[st_out];
            c.ct_crop = refmode ? 0 : (U.k - U.stride) / 2;  // Q1
            c.yg = s2.bu;
            c.g_bs = g_bs;
            c.g_ts = g_ts;
            c.y16 = bul16;
            c.y16_slope = hp.lrelu;
            HIP_OK(conv16("hifigan_upsample_convT", U.up, c, stream, 2.0 * U.up.cin * (double)ssum[st_in] + 6.0 * n_out + (double)U.up.bytes16));
        }
        const bool par = knobs.rb_streams > 1 && nk >= 2 && nk <= 3 && !prof.on;
        if (par) {
            HIP_OK(hipEventRecord(ev_fork_, stream));
            for (size_t j = 1; j < nk; ++j) HIP_OK(hipStreamWaitEvent(side_[j - 1], ev_fork_, 0));
        }
        for (size_t j = 0; j < nk; ++j) {
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            hipStream_t sj = par && j > 0 ? side_[j - 1] : stream;
            const int q = par ? (int)j : 0;
            const Ref16 byl16 = R16(s2.byl[q], C, sts[st_out]), bt16 = R16(s2.bt[q], C, sts[st_out]);
            // narrow stages: each pair runs as ONE kernel and t stays in LDS (rbpair16.hip; bit-identical to the two-kernel path).
            // A fused block reads a halo of its neighbours' input columns while other blocks already write their output, so a fused
            // pair must never write the 16-bit stream it reads: the pairs of a resblock ping-pong between the two 16-bit buffers
            // the two-kernel path uses for the stream and for t. (All pairs of the resblock fuse, or none: a two-kernel pair needs
            // the second buffer for its t.)
            bool fuse_rb = c.fuse16;
            for (size_t d = 0; d < nd; ++d) fuse_rb = fuse_rb && rbpair16_supported(C, R.k, R.dil[d]) && R.c1[d].bias && R.c2[d].bias;
            for (size_t d = 0; d < nd; ++d) {
                const Ref16 in16 = d == 0 ? bul16 : (fuse_rb && (d & 1) == 0 ? bt16 : byl16);
                const Ref16 out16 = fuse_rb && (d & 1) ? bt16 : byl16;  // the stream buffer this pair writes
                Conv16Call c1;
                c1.x = in16;
                c1.len_in = c1.len_out = d_len[st_out];
                c1.batch = B;
                c1.t_in = c1.t_out = smax[st_out];
                c1.sum_in = c1.sum_out = ssum[st_out];
                c1.dil = R.dil[d];
                c1.pad_l = (R.k * R.dil[d] - R.dil[d]) / 2;
                c1.y16 = bt16;  // t = leaky_relu(conv1(...)), rounded: what the second conv consumes (vits.cpp:556-567)
                c1.y16_slope = hp.lrelu;
                const bool fuse = fuse_rb;
                if (!fuse) HIP_OK(conv16("hifigan_resblock_conv1", R.c1[d], c1, sj, 4.0 * n_out + (double)R.c1[d].bytes16));
                Conv16Call c2 = c1;
                c2.x = bt16;
                c2.dil = 1;
                c2.pad_l = (R.k - 1) / 2;
                c2.g_bs = g_bs;
                c2.g_ts = g_ts;
                c2.resg = d == 0 ? s2.bu : s2.by[q];  // residual add (vits.cpp:578), fp32
                c2.y16 = Ref16();
                c2.y16_slope = 1.f;
                double bytes2 = 2.0 * n_out + 4.0 * n_out + 4.0 * n_out + (double)R.c2[d].bytes16;
                if (d + 1 < nd) {
                    c2.yg = s2.by[q];
                    c2.y16 = out16;  // next pair's input
                    c2.y16_slope = hp.lrelu;
                    bytes2 += 2.0 * n_out;
                } else {
                    c2.yg = s2.bs;  // sum over the resblocks and the 1/num_kernels scale (vits.cpp:622-635)
                    if (j > 0) {
                        c2.accg = s2.bs;
                        bytes2 += 4.0 * n_out;
                    }
                    if (j + 1 == nk) {
                        if (refmode) {
                            c2.scale = (float)(1.0 / (double)nk);
                            c2.scale_div = 0;
                        } else {
                            c2.scale = (float)nk;
                            c2.scale_div = 1;
                        }
                        // the stage output is read by the next upsampler (behind leaky_relu, vits.cpp:613) or by conv_post
                        // (behind the final leaky_relu, Q2): its 16-bit copy carries that activation
                        c2.y16 = bsum16;
                        c2.y16_slope = i + 1 < n_up ? hp.lrelu : final_slope;
                        bytes2 += 2.0 * n_out;
                    } else {
                        c2.scale = 1.f;
                    }
                }
                const bool last = d + 1 == nd;
                if (par && last && j > 0) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
                if (fuse) {
                    RbPair16Call f;
                    f.x = c1.x;
                    f.lens = d_len[st_out];
                    f.batch = B;
                    f.tmax = smax[st_out];
                    f.dil = R.dil[d];
                    f.slope = hp.lrelu;
                    f.yg = c2.yg;
                    f.resg = c2.resg;
                    f.accg = c2.accg;
                    f.g_bs = g_bs;
                    f.g_ts = g_ts;
                    f.y16 = c2.y16;
                    f.y16_slope = c2.y16_slope;
                    f.scale = c2.scale;
                    f.scale_div = c2.scale_div;
                    if (prof.on) {
                        char full[160];
                        std::snprintf(full, sizeof(full), "hifigan_resblock_pair|k%d|d%d|F%d|e0g|c%dx%d", R.k, R.dil[d], C, C, C);
                        prof.begin(full, 2.0 * 2.0 * (double)C * C * R.k * (double)ssum[st_out], bytes2 - 2.0 * n_out + (double)R.c1[d].bytes16 + 2.0 * n_out, sj, true);
                    }
                    HIP_OK(launch_rbpair16(R.c1[d], R.c2[d], f, arith_now_, sj));
                    prof.end(sj);
                } else {
                    HIP_OK(conv16("hifigan_resblock_conv2", R.c2[d], c2, sj, bytes2));
                }
                if (par && last) HIP_OK(hipEventRecord(ev_done_[j], sj));
            }
        }
        if (par) HIP_OK(hipStreamWaitEvent(stream, ev_done_[nk - 1], 0));
        cur16 = bsum16;
    }
    prof.begin("hifigan_conv_post_tanh", 2.0 * dec_post_cin_ * dec_post_k_ * (double)ssum[n_up], 2.0 * (dec_post_cin_ + 2) * (double)ssum[n_up], stream);
    HIP_OK(launch_conv_post16(cur16, dec_post_w_, dec_post_cin_, dec_post_k_, pre, wv, d_len[n_up], B, smax[n_up], arith_now_, stream, emit_lo, emit_hi));
    prof.end(stream);
    (void)TR;
    return 0;
}

int Engine::run_vocoder_window32(Call& c, WinCtx& w) {
    std::string& err = c.err;
    const int B = c.B, n_up = c.n_up;
    const int F = hp.flow_size;
    const bool refmode = c.refmode;
    Call::S2& s2 = c.s2;
    const std::vector<int>& sts = c.sts;
    const int lws = c.lws;
    const int Lw = w.Lw;
    const int* const* d_len = w.d_len;
    const std::vector<int>& smax = w.smax;
    const std::vector<int64_t>& ssum = w.ssum;
    const TensorRef zwin = w.zwin, pre = w.pre, wv = w.wv;
    const int emit_lo = w.emit_lo;
    const int* emit_hi = w.emit_hi;
    const float final_slope = w.final_slope;
    auto TR = make_ref;
    TensorRef h0 = TR(s2.h0, hp.up_init, lws);
    {
        ConvCall c;
        c.x = zwin;
        c.y = h0;
        c.len_in = d_len[0];
        c.len_out = d_len[0];
        c.batch = B;
        c.t_in = c.t_out = Lw;
        c.sum_in = c.sum_out = ssum[0];
        c.pad_l = (dec_pre_.kt - 1) / 2;  // padding 3 (vits.cpp:601)
        c.post_act = 2;  // its only reader is the first upsampler, which takes leaky_relu(h0) (vits.cpp:613): activate at the writer
        c.post_slope = hp.lrelu;
        HIP_OK(conv("hifigan_conv_pre", dec_pre_, c));
    }
    TensorRef cur = h0;
    const size_t nk = hp.rb_k.size();
    for (int i = 0; i < n_up; ++i) {
        const UpStageW& U = ups_[i];
        char rx_stage[32];
        std::snprintf(rx_stage, sizeof(rx_stage), "vits.hifigan.stage%d", i);
        RoctxRange rx_stage_range(rx_stage);
        const int C = U.channels, st_in = i, st_out = i + 1;
        TensorRef bu = TR(s2.bu, C, sts[st_out]), bsum = TR(s2.bs, C, sts[st_out]);
        {
            ConvCall c;
            c.x = cur;
            c.y = bu;
            c.len_in = d_len[st_in];
            c.len_out = d_len[st_out];
            c.batch = B;
            c.t_in = smax[st_in];
            c.t_out = smax[st_out];
            c.sum_in = ssum[st_in];
            c.sum_out = ssum[st_out];
            c.pre_act = 0;  // leaky_relu before the upsampler (vits.cpp:613) was applied by whoever wrote `cur`
            c.slope = hp.lrelu;
            c.ct_crop = refmode ? 0 : (U.k - U.stride) / 2;  // Q1 (vits.cpp:187) / HF padding
            if (C >= knobs.lrelu_copy_minc) {  // activated copy for the first conv of each resblock (see below)
                c.y2 = s2.bul;
                c.post_slope = hp.lrelu;
            }
            HIP_OK(conv("hifigan_upsample_convT", U.up, c));
        }
        // resblock j runs on its own stream (engine.h); only the LAST convolution of each resblock touches the shared
        // sum, and those are chained j-1 -> j by events so the additions keep the reference's order (vits.cpp:622-635)
        // (per-kernel event timing needs kernels that do not overlap: the profiler serialises the stage)
        const bool par = knobs.rb_streams > 1 && nk >= 2 && nk <= 3 && !prof.on;
        if (par) {
            HIP_OK(hipEventRecord(ev_fork_, stream));
            for (size_t j = 1; j < nk; ++j) HIP_OK(hipStreamWaitEvent(side_[j - 1], ev_fork_, 0));
        }
        for (size_t j = 0; j < nk; ++j) {
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            hipStream_t sj = par && j > 0 ? side_[j - 1] : stream;
            TensorRef by = TR(s2.by[par ? j : 0], C, sts[st_out]), bt = TR(s2.bt[par ? j : 0], C, sts[st_out]);
            // LeakyReLU is applied where a tensor is WRITTEN, not where it is read: the first conv of a pair stores
            // leaky_relu(t) (t has no other reader), and for wide stages the second conv stores leaky_relu(y) beside y
            // (y itself stays the residual). A reader-side LeakyReLU is VALU work next to the MFMAs — they share the
            // issue port, measured 5 % (k = 11) to 20 % (k = 3) of the K loop — a writer-side one sits in the epilogue.
            const bool lcopy = C >= knobs.lrelu_copy_minc;
            TensorRef byl = TR(s2.byl[par ? j : 0], C, sts[st_out]), bul = TR(s2.bul, C, sts[st_out]);
            // narrow stages: each pair as ONE kernel, t stays in LDS (rbpair32.hip; bit-identical to the two launches below).
            // A fused block reads a halo of its neighbours' input columns while other blocks store their output, so the
            // resblock's stream ping-pongs between `by` and the buffer the two-launch path uses for t. All pairs or none.
            bool fuse_rb = !knobs.no_fuse32;  // (the fused kernel reads the RAW stream: the activated copies of wide stages are for the other resblocks)
            // (16-byte LDS-DMA rows: every buffer a pair may read has to be 16-byte aligned with strides that are multiples of 4)
            auto al16 = [](const TensorRef& t) { return (reinterpret_cast<uintptr_t>(t.p) & 15) == 0 && (t.cs & 3) == 0 && (t.bs & 3) == 0; };
            fuse_rb = fuse_rb && al16(bu) && al16(by) && al16(bt);
            for (size_t d = 0; d < nd && fuse_rb; ++d) fuse_rb = rbpair32_supported(C, R.k, R.dil[d]) && R.c1[d].bias && R.c2[d].bias;
            if (fuse_rb) {
                TensorRef src = bu;
                for (size_t d = 0; d < nd; ++d) {
                    const bool last = d + 1 == nd;
                    RbPair32Call f;
                    f.x = src;
                    f.lens = d_len[st_out];
                    f.batch = B;
                    f.tmax = smax[st_out];
                    f.dil = R.dil[d];
                    f.slope = hp.lrelu;
                    if (!last) {
                        f.y = src.p == by.p ? bt : by;
                    } else {
                        f.y = bsum;  // sum over the resblocks and the 1/num_kernels scale (vits.cpp:622-635), as below
                        if (j > 0) f.acc = bsum;
                        if (j + 1 == nk) {
                            if (refmode) {
                                f.scale = (float)(1.0 / (double)nk);
                                f.scale_div = 0;
                            } else {
                                f.scale = (float)nk;
                                f.scale_div = 1;
                            }
                            if (i + 1 < n_up) {
                                f.post_act = 2;
                                f.post_slope = hp.lrelu;
                            }
                        } else {
                            f.scale = 1.f;
                        }
                    }
                    if (par && last && j > 0) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
                    if (prof.on) {
                        char full[160];
                        std::snprintf(full, sizeof(full), "hifigan_resblock_pair|k%d|d%d|f%d|e0|c%dx%d", R.k, R.dil[d], C, C, C);
                        const double n_out = (double)C * (double)ssum[st_out];
                        prof.begin(full, 2.0 * 2.0 * (double)C * C * R.k * (double)ssum[st_out],
                                   4.0 * n_out * (3 + (f.acc.p ? 1 : 0)) + (double)R.c1[d].bytes + (double)R.c2[d].bytes, sj, true);
                    }
                    HIP_OK(launch_rbpair32(R.c1[d], R.c2[d], f, sj));
                    prof.end(sj);
                    if (par && last) HIP_OK(hipEventRecord(ev_done_[j], sj));
                    src = f.y;
                }
                continue;
            }
            for (size_t d = 0; d < nd; ++d) {
                TensorRef resid = d == 0 ? bu : by;
                ConvCall c1;
                c1.x = lcopy ? (d > 0 ? byl : bul) : resid;
                c1.y = bt;
                c1.len_in = c1.len_out = d_len[st_out];
                c1.batch = B;
                c1.t_in = c1.t_out = smax[st_out];
                c1.sum_in = c1.sum_out = ssum[st_out];
                c1.dil = R.dil[d];
                c1.pad_l = (R.k * R.dil[d] - R.dil[d]) / 2;  // vits.cpp:541-543
                c1.pre_act = lcopy ? 0 : 1;
                c1.slope = hp.lrelu;
                c1.post_act = 2;  // bt = leaky_relu(conv1(...)): what the second conv consumes (vits.cpp:556-566)
                c1.post_slope = hp.lrelu;
                HIP_OK(conv("hifigan_resblock_conv", R.c1[d], c1, sj));
                ConvCall c2 = c1;
                c2.x = bt;
                c2.pre_act = 0;
                c2.post_act = 0;
                c2.y2 = (d + 1 < nd && lcopy) ? byl.p : nullptr;
                c2.dil = 1;
                c2.pad_l = (R.k - 1) / 2;
                c2.res = resid;  // residual add (vits.cpp:578)
                if (d + 1 < nd) c2.y = by;
                else {
                    // last conv of this resblock: fold the sum over resblocks and the 1/num_kernels scale (vits.cpp:622-635)
                    c2.y = bsum;
                    if (j > 0) c2.acc = bsum;
                    if (j + 1 == nk) {
                        if (refmode) {
                            c2.scale = (float)(1.0 / (double)nk);  // ggml_scale by float(1/num_kernels) (vits.cpp:607)
                            c2.scale_div = 0;
                        } else {
                            c2.scale = (float)nk;  // HF divides (modeling_vits.py:546)
                            c2.scale_div = 1;
                        }
                    } else {
                        c2.scale = 1.f;
                    }
                    // (a vocoder with a single resblock kernel has nothing to accumulate: acc stays null and the
                    // scale 1/1 is the identity, so no special case is needed)
                    if (j + 1 == nk && i + 1 < n_up) {
                        // the stage output feeds only the next upsampler, which wants leaky_relu of it (vits.cpp:613);
                        // the last stage stays raw: conv_post applies its own slope (Q2)
                        c2.post_act = 2;
                        c2.post_slope = hp.lrelu;
                    }
                }
                const bool last = d + 1 == nd;
                if (par && last && j > 0) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
                HIP_OK(conv("hifigan_resblock_conv", R.c2[d], c2, sj));
                if (par && last) HIP_OK(hipEventRecord(ev_done_[j], sj));
            }
        }
        if (par) HIP_OK(hipStreamWaitEvent(stream, ev_done_[nk - 1], 0));
        cur = bsum;
    }
    prof.begin("hifigan_conv_post_tanh", 2.0 * dec_post_cin_ * dec_post_k_ * (double)ssum[n_up], 4.0 * (dec_post_cin_ + 1) * (double)ssum[n_up], stream);
    HIP_OK(launch_conv_post(cur, dec_post_w_, dec_post_cin_, dec_post_k_, final_slope, pre, wv, d_len[n_up], B, smax[n_up], stream, emit_lo, emit_hi, arith_now_));
    prof.end(stream);
    (void)F;
    return 0;
}

}  // namespace vits
