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
    Ref16 cur16 = R16(s2.h0, hp.up_init, lws);
    // one to four utterances: conv_pre reads the fp32 flow output itself and runs on conv16_lat_kernel (conv16_lat.hip: no converter launch; same bits)
    const bool pre_lat = !prof.on && F == dec_pre_.cin && conv16_lat_pre_wanted(dec_pre_, B, Lw);
    if (pre_lat) {
        HIP_OK(launch_conv16_lat_pre(dec_pre_, zwin, d_len[0], B, Lw, cur16, hp.lrelu, arith_now_, stream));
    } else {
        prof.begin("to_group16", 0, 6.0 * F * (double)ssum[0], stream, true);
        HIP_OK(launch_to_group16(zwin, d_len[0], B, F, Lw, 1.0f, z16, arith_now_, stream));
        prof.end(stream);
    }
    if (!pre_lat) {
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
        // narrow stages: each RESBLOCK as one kernel (rbblock16.hip: the fp32 stream stays in registers across its three pairs; bit-identical
        // to the pair path). When every resblock of the stage runs that way nobody reads the 16-bit copy of the stage input.
        bool blockrb[3] = {false, false, false};
        bool all_block = nk <= 3;
        for (size_t j = 0; j < nk && j < 3; ++j) {
            const ResBlockW& R = U.rbs[j];
            bool f = c.fuse16 && !knobs.no_rbblock16 && rbblock16_supported(C, R.k, R.dil.data(), (int)R.dil.size(), B, smax[st_out]);
            for (size_t d = 0; d < R.dil.size() && f; ++d) f = R.c1[d].bias && R.c2[d].bias && R.c1[d].wp16 && R.c2[d].wp16;
            blockrb[j] = f;
            all_block = all_block && f;
        }
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
            if (!all_block) {
                c.y16 = bul16;
                c.y16_slope = hp.lrelu;
            }
            const double ct_bytes = 2.0 * U.up.cin * (double)ssum[st_in] + (all_block ? 4.0 : 6.0) * n_out + (double)U.up.bytes16;
            if (convt16_stream_supported(U.up)) {
                // the upsampler as a streaming kernel (convt16.hip: every phase of a tile of input positions in one block; bit-identical)
                if (prof.on) {
                    char full[160], tag[24];
                    convt16_stream_tag(U.up, tag, sizeof(tag));
                    std::snprintf(full, sizeof(full), "hifigan_upsample_convT|k2|d-1|%s|e2g|c%dx%d", tag, U.up.cin, U.up.cout);
                    prof.begin(full, 2.0 * (double)U.up.rows * (double)U.up.cin * 2.0 * (double)ssum[st_in], ct_bytes, stream, true);
                }
                HIP_OK(launch_convt16_stream(U.up, c, arith_now_, stream));
                prof.end(stream);
            } else {
                HIP_OK(conv16("hifigan_upsample_convT", U.up, c, stream, ct_bytes));
            }
        }
        // (small windows — up to four 128-id utterances — run the three resblocks one behind the other: a fork and a join cost ~11 us each per
        // stage, more than the overlap of these short kernels returns: f16 batch 4 2.33 -> 2.18 ms, batch 1 -1 %; from batch 8 on three streams win)
        const bool par = knobs.rb_streams > 1 && nk >= 2 && nk <= 3 && !prof.on && (w.ssum[0] > knobs.rb16_serial_max_frames || w.ssum[0] < knobs.rb16_serial_min_frames);
        // one or two utterances, every resblock of the stage a whole-resblock kernel: the kernels do NOT chain through the shared sum — each writes its own fp32
        // output side by side with the others and launch_rb_sum3 adds them in the reference's order (rbblock16.hip; same bits; the C = 32 stage at batch 1:
        // three chained 15-25 us kernels + two event hand-overs = 105 us, side by side + the sum ~40)
        const bool sum3 = par && (all_block || !knobs.kernel.rb_sum3_block_only) && nk >= 2 && !knobs.kernel.no_rb_sum3 && w.ssum[0] < knobs.rb16_serial_min_frames;
        // Side-by-side whole-resblock kernels (the C = 32 and C = 64 stages, k = 3 / 7 / 11) as ONE launch (rbblock16_group3_kernel) + the sum, on the main stream:
        // no fork, no join (10-30 us of queue hand-over each at batch 1). The members are the kernels' bodies on the same operands: same bits.
        if (sum3 && nk == 3 && c.fuse16 && !knobs.no_rbblock16 && blockrb[0] && blockrb[1]) {
            const int kts[3] = {U.rbs[0].k, U.rbs[1].k, U.rbs[2].k};
            bool dils135 = true;
            for (int m = 0; m < 3; ++m) dils135 = dils135 && U.rbs[m].dil.size() == 3 && U.rbs[m].dil[0] == 1 && U.rbs[m].dil[1] == 3 && U.rbs[m].dil[2] == 5;
            if (dils135 && rbblock16_group3_supported(C, kts, B, smax[st_out])) {
                const PackedConv* w1[3][3];
                const PackedConv* w2[3][3];
                RbBlock16Call f[3];
                for (int m = 0; m < 3; ++m) {
                    for (int d = 0; d < 3; ++d) w1[m][d] = &U.rbs[m].c1[d], w2[m][d] = &U.rbs[m].c2[d];
                    f[m].y0 = s2.bu;
                    f[m].lens = d_len[st_out];
                    f[m].batch = B;
                    f[m].tmax = smax[st_out];
                    f[m].slope = hp.lrelu;
                    f[m].yg = s2.by[m];  // own output: launch_rb_sum3 adds the three in the reference's order, scales, writes the 16-bit copy
                    f[m].g_bs = g_bs;
                    f[m].g_ts = g_ts;
                    f[m].scale = 1.f;
                }
                HIP_OK(launch_rbblock16_group3(w1, w2, f, arith_now_, stream));
                const bool div = !refmode;
                HIP_OK(launch_rb_sum3(s2.by[0], s2.by[1], s2.by[2], C, g_bs, g_ts, d_len[st_out], B, smax[st_out], div ? (float)nk : (float)(1.0 / (double)nk), div ? 1 : 0,
                                      knobs.keep_stage_sum32 ? s2.bs : nullptr, bsum16, i + 1 < n_up ? hp.lrelu : final_slope, arith_now_, stream));
                cur16 = bsum16;
                continue;
            }
        }
        // Side-by-side resblocks whose convs all run on conv16_lat_kernel (the C = 256 stage at one to four utterances): the same-position convs of the three
        // resblocks as ONE launch each (conv16_lat_group_kernel), six launches + the sum on the main stream — no fork, no join (each was 20-45 us of queue
        // hand-over per stage at batch 1). Same kernels' bodies on the same operands: same bits.
        if (sum3 && nk == 3) {
            auto mk16 = [&](size_t j, size_t d, Conv16Call& c1, Conv16Call& c2) {
                const ResBlockW& R = U.rbs[j];
                const size_t nd = R.dil.size();
                const int q = (int)j;
                const Ref16 byl16 = R16(s2.byl[q], C, sts[st_out]), bt16 = R16(s2.bt[q], C, sts[st_out]);
                c1 = Conv16Call();
                c1.x = d == 0 ? bul16 : byl16;
                c1.len_in = c1.len_out = d_len[st_out];
                c1.batch = B;
                c1.t_in = c1.t_out = smax[st_out];
                c1.sum_in = c1.sum_out = ssum[st_out];
                c1.dil = R.dil[d];
                c1.pad_l = (R.k * R.dil[d] - R.dil[d]) / 2;
                c1.y16 = bt16;
                c1.y16_slope = hp.lrelu;
                c2 = c1;
                c2.x = bt16;
                c2.dil = 1;
                c2.pad_l = (R.k - 1) / 2;
                c2.g_bs = g_bs;
                c2.g_ts = g_ts;
                c2.resg = d == 0 ? s2.bu : s2.by[q];
                c2.yg = s2.by[q];  // the stream; behind the last pair the resblock's own output (launch_rb_sum3 adds the three)
                c2.y16 = Ref16();
                c2.y16_slope = 1.f;
                c2.scale = 1.f;
                if (d + 1 < nd) {
                    c2.y16 = byl16;  // next pair's input
                    c2.y16_slope = hp.lrelu;
                }
            };
            bool group = c.fuse16 && !blockrb[0] && !blockrb[1] && !blockrb[2] && U.rbs[0].dil == U.rbs[1].dil && U.rbs[0].dil == U.rbs[2].dil;
            const size_t nd = U.rbs[0].dil.size();
            for (size_t d = 0; d < nd && group; ++d) {
                const PackedConv* w1[3] = {&U.rbs[0].c1[d], &U.rbs[1].c1[d], &U.rbs[2].c1[d]};
                const PackedConv* w2[3] = {&U.rbs[0].c2[d], &U.rbs[1].c2[d], &U.rbs[2].c2[d]};
                Conv16Call a[3], bb[3];
                for (size_t j = 0; j < 3; ++j) mk16(j, d, a[j], bb[j]);
                group = conv16_lat_group_wanted(w1, a) && conv16_lat_group_wanted(w2, bb);
            }
            if (group) {
                for (size_t d = 0; d < nd; ++d) {
                    const PackedConv* w1[3] = {&U.rbs[0].c1[d], &U.rbs[1].c1[d], &U.rbs[2].c1[d]};
                    const PackedConv* w2[3] = {&U.rbs[0].c2[d], &U.rbs[1].c2[d], &U.rbs[2].c2[d]};
                    Conv16Call a[3], bb[3];
                    for (size_t j = 0; j < 3; ++j) mk16(j, d, a[j], bb[j]);
                    HIP_OK(launch_conv16_lat_group(w1, a, arith_now_, stream));
                    HIP_OK(launch_conv16_lat_group(w2, bb, arith_now_, stream));
                }
                const bool div = !refmode;
                HIP_OK(launch_rb_sum3(s2.by[0], s2.by[1], s2.by[2], C, g_bs, g_ts, d_len[st_out], B, smax[st_out], div ? (float)nk : (float)(1.0 / (double)nk), div ? 1 : 0,
                                      knobs.keep_stage_sum32 ? s2.bs : nullptr, bsum16, i + 1 < n_up ? hp.lrelu : final_slope, arith_now_, stream));
                cur16 = bsum16;
                continue;
            }
        }
        if (par) {
            HIP_OK(hipEventRecord(ev_fork_, stream));
            for (size_t j = 1; j < nk; ++j) HIP_OK(hipStreamWaitEvent(side_[j - 1], ev_fork_, 0));
        }
        // (side-by-side resblocks — sum3 — need no order: the LAST one, the longest chain (k = 11), is enqueued first and on the main stream, where it starts
        // without the fork's cross-queue hand-over (10-40 us later on the side streams at batch 1); the short k = 3 chain takes the last side stream)
        const bool longest_first = sum3 && !knobs.kernel.rb_sum3_in_order;
        for (size_t jj = 0; jj < nk; ++jj) {
            const size_t j = longest_first ? nk - 1 - jj : jj;
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            hipStream_t sj = par && jj > 0 ? side_[jj - 1] : stream;
            const int q = par ? (int)j : 0;
            const Ref16 byl16 = R16(s2.byl[q], C, sts[st_out]), bt16 = R16(s2.bt[q], C, sts[st_out]);
            // narrow stages: each pair runs as ONE kernel and t stays in LDS (rbpair16.hip; bit-identical to the two-kernel path).
            // A fused block reads a halo of its neighbours' input columns while other blocks already write their output, so a fused
            // pair must never write the 16-bit stream it reads: the pairs of a resblock ping-pong between the two 16-bit buffers
            // the two-kernel path uses for the stream and for t. (All pairs of the resblock fuse, or none: a two-kernel pair needs
            // the second buffer for its t.)
            if (j < 3 && blockrb[j]) {
                const PackedConv* w1[3] = {&R.c1[0], &R.c1[1], &R.c1[2]};
                const PackedConv* w2[3] = {&R.c2[0], &R.c2[1], &R.c2[2]};
                RbBlock16Call f;
                f.y0 = s2.bu;
                f.lens = d_len[st_out];
                f.batch = B;
                f.tmax = smax[st_out];
                f.slope = hp.lrelu;
                f.yg = s2.bs;  // sum over the resblocks and the 1/num_kernels scale (vits.cpp:622-635), as the last pair of the pair path
                f.g_bs = g_bs;
                f.g_ts = g_ts;
                double bytes = (4.0 + 4.0) * n_out;
                if (j > 0) {
                    f.accg = s2.bs;
                    bytes += 4.0 * n_out;
                }
                if (j + 1 == nk) {
                    if (refmode) {
                        f.scale = (float)(1.0 / (double)nk);
                        f.scale_div = 0;
                    } else {
                        f.scale = (float)nk;
                        f.scale_div = 1;
                    }
                    f.y16 = bsum16;
                    f.y16_slope = i + 1 < n_up ? hp.lrelu : final_slope;
                    // the stage output has ONE reader — the next upsampler or conv_post, through the 16-bit copy: the fp32 sum of the last
                    // resblock is a dead store (4 of its 10-14 bytes per element: 1.5 GB per batch of 64 x 128 ids over the four stages)
                    if (!knobs.keep_stage_sum32) {
                        f.yg = nullptr;
                        bytes -= 4.0 * n_out;
                    }
                    bytes += 2.0 * n_out;
                }
                if (sum3) {  // own output, no accumulation, no scale, no 16-bit copy: launch_rb_sum3 below does those
                    f.yg = s2.by[q];
                    f.accg = nullptr;
                    f.scale = 1.f;
                    f.scale_div = 0;
                    f.y16 = Ref16();
                } else if (par && j > 0)
                    HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));  // (the accumulation is inside the kernel: the resblocks chain)
                if (prof.on) {
                    char full[160];
                    std::snprintf(full, sizeof(full), "hifigan_resblock_block|k%d|d135|B%d|e0g|c%dx%d", R.k, C, C, C);
                    for (size_t d = 0; d < nd; ++d) bytes += (double)R.c1[d].bytes16 + (double)R.c2[d].bytes16;
                    prof.begin(full, 3.0 * 2.0 * 2.0 * (double)C * C * R.k * (double)ssum[st_out], bytes, sj, true);
                }
                HIP_OK(launch_rbblock16(w1, w2, f, arith_now_, sj));
                prof.end(sj);
                if (par) HIP_OK(hipEventRecord(ev_done_[j], sj));
                continue;
            }
            bool fuse_rb = c.fuse16;
            for (size_t d = 0; d < nd; ++d) fuse_rb = fuse_rb && rbpair16_supported(C, R.k, R.dil[d]) && R.c1[d].bias && R.c2[d].bias;
            // one or a few utterances on a wide stage: a fused pair is 28-32 blocks that each stream both convs' weights through one CU; two launches of
            // conv16_lat_kernel deal the rows out over the chip (conv16_lat.hip; same bits)
            if (fuse_rb && C >= 128) {
                bool lat = true;
                for (size_t d = 0; d < nd; ++d) lat = lat && conv16_lat_shape_ok(C, R.k, R.dil[d], B, smax[st_out]);
                if (lat) fuse_rb = false;
            }
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
                } else if (sum3) {
                    c2.yg = s2.by[q];  // the resblock's own output; launch_rb_sum3 below adds the three in the reference's order, scales, writes the 16-bit copy
                    c2.scale = 1.f;
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
                        if (!knobs.keep_stage_sum32) {  // (dead store: see the whole-resblock path above)
                            c2.yg = nullptr;
                            bytes2 -= 4.0 * n_out;
                        }
                    } else {
                        c2.scale = 1.f;
                    }
                }
                const bool last = d + 1 == nd;
                if (par && last && j > 0 && !sum3) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
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
        if (sum3) {
            for (size_t j = 0; j + 1 < nk; ++j) HIP_OK(hipStreamWaitEvent(stream, ev_done_[j], 0));  // (every chain, not only the last: they no longer wait for each other)
            const bool div = !refmode;
            prof.begin("hifigan_resblock_sum", 0, (4.0 * nk + 2.0) * n_out, stream);
            HIP_OK(launch_rb_sum3(s2.by[0], s2.by[1], nk > 2 ? s2.by[2] : nullptr, C, g_bs, g_ts, d_len[st_out], B, smax[st_out], div ? (float)nk : (float)(1.0 / (double)nk), div ? 1 : 0,
                                  knobs.keep_stage_sum32 ? s2.bs : nullptr, bsum16, i + 1 < n_up ? hp.lrelu : final_slope, arith_now_, stream));
            prof.end(stream);
        }
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
        // ---- the resblocks of this stage (vits.cpp:622-635): independent chains of `nd` conv pairs on the same input that meet only in
        // the sum. Only the LAST launch of each chain touches the shared sum, and those run in the reference's order (RB0, += RB1,
        // += RB2 and the 1/num_kernels scale) on the main stream. Everything before them is scheduled one of two ways:
        //   grouped  — the same-position convs of the resblocks as ONE launch (conv_group_kernel: 11-tap blocks first, 3-tap blocks
        //              last, one grid tail instead of three); resblocks that run as fused pairs (rbpair32) keep their own chain;
        //   separate — every resblock its own chain of launches, on three streams (or serialised under the profiler, whose
        //              per-kernel events need kernels that do not overlap).
        // Same kernels bodies, same operands, same order of the additions either way: the PCM is bit-identical (GPU test).
        const bool lcopy = C >= knobs.lrelu_copy_minc;
        TensorRef bul = TR(s2.bul, C, sts[st_out]);
        auto al16 = [](const TensorRef& t) { return (reinterpret_cast<uintptr_t>(t.p) & 15) == 0 && (t.cs & 3) == 0 && (t.bs & 3) == 0; };
        // which resblocks run as fused pairs (narrow stages; all pairs of a resblock or none). rbpair32 and the grouped launch are fp32
        // kernels that do not go through conv(): in a 16-bit arithmetic mode on this (converter) path they would silently compute the
        // wide-stage resblocks in fp32 — not the arithmetic that was asked for, and other bits with the profiler on than off.
        const bool exact32 = arith_now_ == VITS_ARITH_F32;
        bool fusedrb[3] = {false, false, false};
        for (size_t j = 0; j < nk && j < 3; ++j) {
            const ResBlockW& R = U.rbs[j];
            bool f = exact32 && !knobs.no_fuse32 && al16(bu) && (reinterpret_cast<uintptr_t>(s2.by[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(s2.bt[0]) & 15) == 0 && (sts[st_out] & 3) == 0;
            for (size_t d = 0; d < R.dil.size() && f; ++d) f = rbpair32_supported(C, R.k, R.dil[d]) && R.c1[d].bias && R.c2[d].bias;
            // VITS_ARITH_F32_SPLIT: a resblock the split kernels take (conv_split.hip: C >= 128, any tap count) runs un-fused — the C = 128, k = 3 pairs, fused in
            // the exact mode, are 1.56 ms each there and 0.9 as two split convs
            if (f && split_on() && s2.sp_u && C >= 128) {
                bool sp = true;
                for (size_t d = 0; d < R.dil.size() && sp; ++d) sp = conv_split_supported(R.c1[d], R.dil[d]) && conv_split_supported(R.c2[d], 1);
                if (sp) f = false;
            }
            fusedrb[j] = f;
        }
        // narrow stages, 3-tap resblocks: the WHOLE resblock as one kernel (rbblock32.hip: the stream stays in registers across its three pairs;
        // bit-identical to the fused pairs). It carries the accumulation into the shared sum itself, so it is launched where the resblock's last
        // launch would be.
        bool blockrb[3] = {false, false, false};
        for (size_t j = 0; j < nk && j < 3; ++j) {
            const ResBlockW& R = U.rbs[j];
            bool f = fusedrb[j] && !knobs.no_rbblock32 && rbblock32_supported(C, R.k, R.dil.data(), (int)R.dil.size());
            for (size_t d = 0; d < R.dil.size() && f; ++d) f = R.c1[d].wp && R.c2[d].wp;
            blockrb[j] = f;
        }
        // VITS_ARITH_F32_SPLIT: the un-fused resblocks of a wide stage run their convs on the bf16 matrix cores with split operands (conv_split.hip): their
        // inputs are three-plane tensors written by the producing epilogue — the stage input by a converter launch —, the fp32 stream and the sum stay put.
        bool splitrb[3] = {false, false, false};
        bool any_split = false;
        for (size_t j = 0; j < nk && j < 3 && split_on() && s2.sp_u; ++j) {
            const ResBlockW& R = U.rbs[j];
            bool f = !fusedrb[j] && C >= 128;
            for (size_t d = 0; d < R.dil.size() && f; ++d) f = conv_split_supported(R.c1[d], R.dil[d]) && conv_split_supported(R.c2[d], 1);
            splitrb[j] = f;
            any_split = any_split || f;
        }
        auto SR = [&](uint16_t* ptr) {
            Split3Ref r;
            r.p = ptr;
            r.ts = sts[st_out];
            r.ps = (int64_t)C * sts[st_out];
            r.bs = 3 * r.ps;
            return r;
        };
        if (any_split) {
            prof.begin("split_planes", 0, 10.0 * (double)C * (double)ssum[st_out], stream);
            HIP_OK(launch_split_planes(bu, C, d_len[st_out], B, smax[st_out], hp.lrelu, SR(s2.sp_u), stream));
            prof.end(stream);
        }
        // (small grids, set below: every resblock writes its own output — s2.by[j] — and launch_rb_sum3_std adds them in the reference's order: no resblock's last
        // launch waits for the previous resblock's; same bits)
        bool sum3 = false;
        auto run_block = [&](size_t j, hipStream_t sj) -> int {
            const ResBlockW& R = U.rbs[j];
            const PackedConv* w1[3] = {&R.c1[0], &R.c1[1], &R.c1[2]};
            const PackedConv* w2[3] = {&R.c2[0], &R.c2[1], &R.c2[2]};
            RbBlock32Call f;
            f.x = bu;
            f.lens = d_len[st_out];
            f.batch = B;
            f.tmax = smax[st_out];
            f.slope = hp.lrelu;
            f.y = bsum;  // sum over the resblocks and the 1/num_kernels scale (vits.cpp:622-635), as the last pair of the pair path
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
            if (sum3) {
                f.y = TR(s2.by[j], C, sts[st_out]);
                f.acc = TensorRef();
                f.scale = 1.f;
                f.scale_div = 0;
                f.post_act = 0;
            }
            if (prof.on) {
                char full[160];
                std::snprintf(full, sizeof(full), "hifigan_resblock_block|k%d|d135|b%d|e0|c%dx%d", R.k, C, C, C);
                const double n_out = (double)C * (double)ssum[st_out];
                double bytes = 4.0 * n_out * (2 + (f.acc.p ? 1 : 0));
                for (size_t d = 0; d < R.dil.size(); ++d) bytes += (double)R.c1[d].bytes + (double)R.c2[d].bytes;
                prof.begin(full, 3.0 * 2.0 * 2.0 * (double)C * C * R.k * (double)ssum[st_out], bytes, sj, true);
            }
            HIP_OK(launch_rbblock32(w1, w2, f, sj));
            prof.end(sj);
            return 0;
        };
        // grouped schedule: at least two un-fused resblocks with distinct tap counts of {11, 7, 3}, the same dilation list, on the 128 x 128 tile
        // (measured, batch 64 x 128 ids: serialised launches 79.7 ms per step, grouped 79.1, three streams 76.8 — kernels of DIFFERENT
        // launches share a CU, which blocks of one launch do not, so the streams win where they can be used: the
        // grouped schedule is for the single-stream case, i.e. under the per-kernel profiler; VITS_RB_GROUP=1 forces it)
        bool grouped = exact32 && !any_split && !knobs.no_rb_group && nk >= 2 && nk <= 3 && knobs.rb_streams > 1 && (prof.on || knobs.rb_group_always);
        {
            int members = 0, seen = 0;
            for (size_t j = 0; j < nk && grouped; ++j) {
                const ResBlockW& R = U.rbs[j];
                if (R.dil != U.rbs[0].dil) grouped = false;
                if (fusedrb[j]) continue;
                const int bit = R.k == 11 ? 1 : R.k == 7 ? 2 : R.k == 3 ? 4 : 0;
                if (!bit || (seen & bit)) grouped = false;
                seen |= bit;
                ConvCall shape;  // (what the tile rule looks at: small grids step down from the 128 x 128 tile and are not grouped)
                shape.batch = B;
                shape.t_in = shape.t_out = smax[st_out];
                for (size_t d = 0; d < R.dil.size() && grouped; ++d)
                    grouped = conv_group_supported(R.c1[d], R.dil[d]) && conv_group_supported(R.c2[d], 1) && resolve_conv_tile(R.c1[d], shape) == TILE_128x128;
                ++members;
            }
            grouped = grouped && members >= 2;
        }
        const bool par = knobs.rb_streams > 1 && nk >= 2 && nk <= 3 && !prof.on;
        auto bufq = [&](size_t j) { return (par || grouped) ? (int)j : 0; };  // resblocks that overlap in time need their own (y, t, y') buffers
        // conv 1 / conv 2 of pair d of resblock j (two-launch form)
        auto mk_c1 = [&](size_t j, size_t d) {
            const ResBlockW& R = U.rbs[j];
            const int q = bufq(j);
            TensorRef by = TR(s2.by[q], C, sts[st_out]), bt = TR(s2.bt[q], C, sts[st_out]), byl = TR(s2.byl[q], C, sts[st_out]);
            TensorRef resid = d == 0 ? bu : by;
            // LeakyReLU is applied where a tensor is WRITTEN, not where it is read: the first conv of a pair stores leaky_relu(t)
            // (t has no other reader), and for wide stages the second conv stores leaky_relu(y) beside y (y itself stays the
            // residual). A reader-side LeakyReLU is VALU work next to the MFMAs — they share the issue port, measured 5 % (k = 11)
            // to 20 % (k = 3) of the K loop — a writer-side one sits in the epilogue.
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
            if (j < 3 && splitrb[j]) {
                // split arithmetic: the input is the planes of leaky_relu(stage input / stream), the output ONLY the planes of leaky_relu(t)
                c1.x = TensorRef();
                c1.xs3 = d > 0 ? SR(s2.sp_y[q]) : SR(s2.sp_u);
                c1.y = TensorRef();
                c1.ys3 = SR(s2.sp_t[q]);
                c1.ys3_slope = hp.lrelu;
                c1.pre_act = 0;
                c1.post_act = 0;
            }
            return c1;
        };
        auto mk_c2 = [&](size_t j, size_t d) {
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            const int q = bufq(j);
            TensorRef by = TR(s2.by[q], C, sts[st_out]), bt = TR(s2.bt[q], C, sts[st_out]), byl = TR(s2.byl[q], C, sts[st_out]);
            ConvCall c2 = mk_c1(j, d);
            c2.x = bt;
            c2.pre_act = 0;
            c2.post_act = 0;
            c2.y2 = (d + 1 < nd && lcopy) ? byl.p : nullptr;
            c2.dil = 1;
            c2.pad_l = (R.k - 1) / 2;
            c2.res = d == 0 ? bu : by;  // residual add (vits.cpp:578)
            if (j < 3 && splitrb[j]) {
                c2.y = by;  // (mk_c1 cleared it; the last conv of the resblock redirects it to the sum below)
                c2.xs3 = SR(s2.sp_t[q]);
                c2.y2 = nullptr;
                c2.ys3 = Split3Ref();
                if (d + 1 < nd) {
                    c2.ys3 = SR(s2.sp_y[q]);  // planes of leaky_relu(y'): the next pair's first conv
                    c2.ys3_slope = hp.lrelu;
                }
            }
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
                // (a vocoder with a single resblock kernel has nothing to accumulate: acc stays null and the scale 1/1 is the
                // identity, so no special case is needed)
                if (j + 1 == nk && i + 1 < n_up) {
                    // the stage output feeds only the next upsampler, which wants leaky_relu of it (vits.cpp:613); the last
                    // stage stays raw: conv_post applies its own slope (Q2)
                    c2.post_act = 2;
                    c2.post_slope = hp.lrelu;
                }
                if (sum3) {
                    c2.y = by;
                    c2.acc = TensorRef();
                    c2.scale = 1.f;
                    c2.scale_div = 0;
                    c2.post_act = 0;
                }
            }
            return c2;
        };
        // pair d of resblock j as ONE kernel, t stays in LDS (rbpair32.hip; bit-identical to the two launches). A fused block reads a
        // halo of its neighbours' input columns while other blocks store their output, so the resblock's stream ping-pongs between
        // `by` and the buffer the two-launch path uses for t.
        auto run_fused = [&](size_t j, size_t d, hipStream_t sj) -> int {
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            const int q = bufq(j);
            TensorRef by = TR(s2.by[q], C, sts[st_out]), bt = TR(s2.bt[q], C, sts[st_out]);
            const bool last = d + 1 == nd;
            RbPair32Call f;
            f.x = d == 0 ? bu : ((d & 1) ? by : bt);
            f.lens = d_len[st_out];
            f.batch = B;
            f.tmax = smax[st_out];
            f.dil = R.dil[d];
            f.slope = hp.lrelu;
            if (!last) {
                f.y = (d & 1) ? bt : by;
            } else {
                f.y = bsum;  // sum over the resblocks and the 1/num_kernels scale (vits.cpp:622-635), as in mk_c2
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
                if (sum3) {
                    f.y = by;  // (the last pair's input is bt: nd is odd)
                    f.acc = TensorRef();
                    f.scale = 1.f;
                    f.scale_div = 0;
                    f.post_act = 0;
                }
            }
            if (prof.on) {
                char full[160];
                std::snprintf(full, sizeof(full), "hifigan_resblock_pair|k%d|d%d|f%d|e0|c%dx%d", R.k, R.dil[d], C, C, C);
                const double n_out = (double)C * (double)ssum[st_out];
                prof.begin(full, 2.0 * 2.0 * (double)C * C * R.k * (double)ssum[st_out], 4.0 * n_out * (3 + (f.acc.p ? 1 : 0)) + (double)R.c1[d].bytes + (double)R.c2[d].bytes, sj, true);
            }
            HIP_OK(launch_rbpair32(R.c1[d], R.c2[d], f, sj));
            prof.end(sj);
            return 0;
        };
        if (grouped) {
            // ---- grouped schedule ---------------------------------------------------------------------------------------------------
            const size_t nd = U.rbs[0].dil.size();
            // fused resblocks: their chain up to (not including) the last pair, beside the group — on a side stream unless the profiler
            // needs kernels that do not overlap
            bool any_fused = false;
            for (size_t j = 0; j < nk; ++j) any_fused = any_fused || (fusedrb[j] && !blockrb[j]);
            hipStream_t sf = (any_fused && !prof.on) ? side_[0] : stream;
            if (sf != stream) {
                HIP_OK(hipEventRecord(ev_fork_, stream));
                HIP_OK(hipStreamWaitEvent(sf, ev_fork_, 0));
            }
            for (size_t j = 0; j < nk; ++j)
                if (fusedrb[j] && !blockrb[j])
                    for (size_t d = 0; d + 1 < nd; ++d)
                        if (run_fused(j, d, sf)) return -1;
            if (sf != stream) HIP_OK(hipEventRecord(ev_done_[0], sf));
            auto run_group = [&](size_t d, bool second) -> int {
                const PackedConv* gw[3];
                ConvCall gc[3];
                int n = 0;
                double flop = 0, bytes = 0;
                for (size_t j = 0; j < nk; ++j) {
                    if (fusedrb[j]) continue;
                    const ResBlockW& R = U.rbs[j];
                    gw[n] = second ? &R.c2[d] : &R.c1[d];
                    gc[n] = second ? mk_c2(j, d) : mk_c1(j, d);
                    const ConvCall& cc = gc[n];
                    flop += conv_flops(*gw[n], cc, ssum[st_out]);
                    bytes += 4.0 * ((double)C * ssum[st_out] * (2 + (cc.res.p ? 1 : 0) + (cc.acc.p ? 1 : 0) + (cc.y2 ? 1 : 0))) + (double)gw[n]->bytes;
                    ++n;
                }
                if (prof.on) {
                    char full[160];
                    std::snprintf(full, sizeof(full), "hifigan_resblock_group%d|kG|d%d|G0|e0|c%dx%d", n, second ? 1 : U.rbs[0].dil[d], C, C);
                    prof.begin(full, flop, bytes, stream, /*chain=*/true);
                }
                HIP_OK(launch_conv_group(gw, gc, n, stream));
                prof.end(stream);
                return 0;
            };
            for (size_t d = 0; d < nd; ++d) {
                if (run_group(d, false)) return -1;
                if (d + 1 < nd && run_group(d, true)) return -1;
            }
            // the last launch of every resblock, in the reference's order of the additions
            if (sf != stream) HIP_OK(hipStreamWaitEvent(stream, ev_done_[0], 0));
            for (size_t j = 0; j < nk; ++j) {
                if (blockrb[j]) {
                    if (run_block(j, stream)) return -1;
                } else if (fusedrb[j]) {
                    if (run_fused(j, nd - 1, stream)) return -1;
                } else {
                    HIP_OK(conv("hifigan_resblock_conv", U.rbs[j].c2[nd - 1], mk_c2(j, nd - 1), stream));
                }
            }
            cur = bsum;
            continue;
        }
        // ---- separate schedule: resblock j on its own stream (engine.h), the last launches chained j-1 -> j by events ---------------
        // up to eight 128-id utterances: side-by-side resblocks (see `sum3` above), the last — longest — chain enqueued first and on the main stream (as the 16-bit path)
        {
            bool odd = true;
            for (size_t j = 0; j < nk; ++j) odd = odd && (U.rbs[j].dil.size() & 1);
            sum3 = par && odd && !any_split && !knobs.kernel.no_rb_sum3 && !knobs.kernel.no_rb_sum3_f32 && w.ssum[0] < knobs.rb32_sum3_max_frames;
        }
        if (par) {
            HIP_OK(hipEventRecord(ev_fork_, stream));
            for (size_t j = 1; j < nk; ++j) HIP_OK(hipStreamWaitEvent(side_[j - 1], ev_fork_, 0));
        }
        for (size_t jj = 0; jj < nk; ++jj) {
            const size_t j = sum3 ? nk - 1 - jj : jj;
            const ResBlockW& R = U.rbs[j];
            const size_t nd = R.dil.size();
            hipStream_t sj = par && jj > 0 ? side_[jj - 1] : stream;
            const bool fuse_rb = j < 3 ? fusedrb[j] : false;
            if (j < 3 && blockrb[j]) {
                // (the kernel adds into the shared sum: it takes the place of the resblock's last launch in the chain of additions)
                if (par && j > 0 && !sum3) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
                if (run_block(j, sj)) return -1;
                if (par) HIP_OK(hipEventRecord(ev_done_[j], sj));
                continue;
            }
            for (size_t d = 0; d < nd; ++d) {
                const bool last = d + 1 == nd;
                if (!fuse_rb) HIP_OK(conv("hifigan_resblock_conv", R.c1[d], mk_c1(j, d), sj));
                if (par && last && j > 0 && !sum3) HIP_OK(hipStreamWaitEvent(sj, ev_done_[j - 1], 0));
                if (fuse_rb) {
                    if (run_fused(j, d, sj)) return -1;
                } else {
                    HIP_OK(conv("hifigan_resblock_conv", R.c2[d], mk_c2(j, d), sj));
                }
                if (par && last) HIP_OK(hipEventRecord(ev_done_[j], sj));
            }
        }
        if (par) HIP_OK(hipStreamWaitEvent(stream, ev_done_[nk - 1], 0));
        if (sum3) {
            for (size_t j = 0; j + 1 < nk; ++j) HIP_OK(hipStreamWaitEvent(stream, ev_done_[j], 0));
            const bool div = !refmode;
            const bool act = i + 1 < n_up;  // (the next upsampler wants leaky_relu of the stage output, vits.cpp:613: where the last resblock's epilogue applied it)
            prof.begin("hifigan_resblock_sum", 0, 4.0 * (nk + 1) * (double)C * (double)ssum[st_out], stream);
            HIP_OK(launch_rb_sum3_std(TR(s2.by[0], C, sts[st_out]), TR(s2.by[1], C, sts[st_out]), nk > 2 ? TR(s2.by[2], C, sts[st_out]) : TensorRef(), bsum, C, d_len[st_out], B, smax[st_out],
                                      div ? (float)nk : (float)(1.0 / (double)nk), div ? 1 : 0, act ? 2 : 0, hp.lrelu, stream));
            prof.end(stream);
        }
        cur = bsum;
    }
    prof.begin("hifigan_conv_post_tanh", 2.0 * dec_post_cin_ * dec_post_k_ * (double)ssum[n_up], 4.0 * (dec_post_cin_ + 1) * (double)ssum[n_up], stream);
    HIP_OK(launch_conv_post(cur, dec_post_w_, dec_post_cin_, dec_post_k_, final_slope, pre, wv, d_len[n_up], B, smax[n_up], stream, emit_lo, emit_hi, arith_now_));
    prof.end(stream);
    (void)F;
    return 0;
}

}  // namespace vits
