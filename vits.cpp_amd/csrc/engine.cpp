// engine.cpp — one call from ids to PCM (see engine.h): input checks, the two arenas, the single host read of the frame counts,
// vocoder windows / streaming, results. The phases themselves live in engine_stage1.cpp, engine_flow.cpp and engine_vocoder.cpp.
// Compiled with hipcc as host C++.
#include "engine_internal.h"

namespace vits {

// ---- convolution wrappers (profiled launches) -------------------------------------------------------------------------------
// One 16-bit-operand convolution launch (conv16.hip), profiled like conv(): tile names carry a capital T
hipError_t Engine::conv16(const char* name, const PackedConv& w, const Conv16Call& c, hipStream_t stream, double bytes) {
    if (prof.on) {
        const int ncols = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
        const int tile = c.tile >= 0 ? c.tile : (conv16_lat_wanted(w, c) ? 7 : choose_conv16_tile(w.rows, w.epi, ncols, w.mtiles_used, c.batch));
        const bool group = c.yg || c.y16.p;
        char full[160];
        std::snprintf(full, sizeof(full), "%s|k%d|d%d|T%d|e%d%s|c%dx%d", name, w.kt, w.epi == EPI_CONVT ? -1 : (w.kt == 1 ? 1 : c.dil), tile, w.epi, group ? "g" : "", w.cin,
                      w.cout);
        const int64_t tot_in = c.sum_in >= 0 ? c.sum_in : (int64_t)c.batch * c.t_in;
        const int64_t tot_out = c.sum_out >= 0 ? c.sum_out : (int64_t)c.batch * c.t_out;
        const int64_t cols = w.epi == EPI_CONVT ? tot_in : tot_out;
        prof.begin(full, 2.0 * (double)w.rows * (double)w.cin * (double)w.kt * (double)cols, bytes, stream, /*chain=*/true);
    }
    hipError_t e = launch_conv16(w, c, arith_now_, stream);
    prof.end(stream);
    return e;
}

// A conv of the fp32-layout orchestration in a 16-bit arithmetic mode: (1) the input is rounded into the 16-bit group layout,
// with the conv's input leaky_relu fused (the reference's leaky_relu node + fp16 im2col, vits.cpp:554 + custom-ops.h:684-690),
// (2) the 16-bit-operand kernel writes the same fp32 outputs the fp32 kernel would. Used for stage one, the flow, and any
// vocoder whose channel counts rule out the group-layout fast path.
hipError_t Engine::conv16_transparent(const char* name, const PackedConv& w, const ConvCall& c, hipStream_t stream) {
    int si = 0;
    if (stream == side_[0]) si = 1;
    else if (stream == side_[1]) si = 2;
    Ref16 x16 = x16_[si];
    const int groups = (w.cin + 7) / 8;
    x16.ts = round_up(std::max(c.t_in, 1), 8);
    x16.bs = (int64_t)groups * x16.ts * 8;
    if (!x16.p || (size_t)c.batch * (size_t)x16.bs > x16_cap_[si]) return hipErrorOutOfMemory;
    const int64_t tot_in = c.sum_in >= 0 ? c.sum_in : (int64_t)c.batch * c.t_in;
    const int64_t tot_out = c.sum_out >= 0 ? c.sum_out : (int64_t)c.batch * c.t_out;
    prof.begin("to_group16", 0, 6.0 * (double)w.cin * (double)tot_in, stream, true);
    hipError_t e = launch_to_group16(c.x, c.len_in, c.batch, w.cin, c.t_in, c.pre_act ? c.slope : 1.0f, x16, arith_now_, stream);
    prof.end(stream);
    if (e != hipSuccess) return e;
    Conv16Call k;
    k.x = x16;
    k.len_in = c.len_in;
    k.len_out = c.len_out;
    k.batch = c.batch;
    k.t_in = c.t_in;
    k.t_out = c.t_out;
    k.dil = c.dil;
    k.pad_l = c.pad_l;
    k.post_act = c.post_act;
    k.post_slope = c.post_slope;
    k.scale = c.scale;
    k.scale_div = c.scale_div;
    k.ct_crop = c.ct_crop;
    k.y = c.y;
    k.res = c.res;
    k.acc = c.acc;
    k.y2 = c.y2;
    k.sum_in = c.sum_in;
    k.sum_out = c.sum_out;
    const int cout_stored = w.epi == EPI_GATE ? w.cout / 2 : w.cout;
    const double bytes = 2.0 * (double)w.cin * tot_in + 4.0 * (double)cout_stored * tot_out * (1 + (c.res.p ? 1 : 0) + (c.acc.p ? 1 : 0) + (c.y2 ? 1 : 0)) + (double)w.bytes16;
    return conv16(name, w, k, stream, bytes);
}

hipError_t Engine::conv(const char* name, const PackedConv& w, ConvCall c, hipStream_t on) {
    hipStream_t stream = on ? on : this->stream;
    if (c.xs3.p) {
        // VITS_ARITH_F32_SPLIT: a conv whose input exists as split planes (the wide stages' resblocks, engine_vocoder.cpp) runs on conv_split.hip
        if (prof.on) {
            char full[160];
            std::snprintf(full, sizeof(full), "%s|k%d|d%d|S128|e0|c%dx%d", name, w.kt, c.dil, w.cin, w.cout);
            const int64_t tot = c.sum_out >= 0 ? c.sum_out : (int64_t)c.batch * c.t_out;
            const double bytes = (double)tot * (6.0 * w.cin + (double)w.cout * (4.0 * ((c.y.p ? 1 : 0) + (c.res.p ? 1 : 0) + (c.acc.p ? 1 : 0)) + (c.ys3.p ? 6.0 : 0.0))) + (double)w.bytes_s;
            prof.begin(full, conv_flops(w, c, tot), bytes, stream, /*chain=*/true);
        }
        hipError_t e = launch_conv_split(w, c, stream);
        prof.end(stream);
        return e;
    }
    if (arith_now_ != VITS_ARITH_F32 && w.wp16) return conv16_transparent(name, w, c, stream);
    if (prof.on) {
        // name = label|k<taps>|d<dilation>|t<tile>|e<epilogue>|c<cin>x<cout>: one entry per kernel instantiation and shape, so the
        // bench can line entries up with rocprofv3's per-kernel-name statistics
        const int tile = resolve_conv_tile(w, c);
        char full[160];
        std::snprintf(full, sizeof(full), "%s|k%d|d%d|t%d|e%d|c%dx%d", name, w.kt, w.epi == EPI_CONVT ? -1 : (w.kt == 1 ? 1 : c.dil), tile, w.epi, w.cin,
                      w.cout);
        // algorithmic work over the REAL lengths (sum over utterances), not the padded grid extent
        const int64_t tot_in = c.sum_in >= 0 ? c.sum_in : (int64_t)c.batch * c.t_in;
        const int64_t tot_out = c.sum_out >= 0 ? c.sum_out : (int64_t)c.batch * c.t_out;
        const int64_t cols = w.epi == EPI_CONVT ? tot_in : tot_out;
        // algorithmic bytes: input read once, output (and its activated copy, if any) written once, residual/accumulator read once, weights once
        const int cout_stored = w.epi == EPI_GATE ? w.cout / 2 : w.cout;
        double bytes = 4.0 * ((double)w.cin * tot_in + (double)cout_stored * tot_out * (1 + (c.res.p ? 1 : 0) + (c.acc.p ? 1 : 0) + (c.y2 ? 1 : 0))) + (double)w.bytes;
        prof.begin(full, conv_flops(w, c, cols), bytes, stream, /*chain=*/true);
    }
    hipError_t e = launch_conv(w, c, stream);
    prof.end(stream);
    return e;
}

int Engine::sync(std::string& err) {
    if (front_) HIP_OK(hipStreamSynchronize(front_));
    HIP_OK(hipStreamSynchronize(stream));
    async_tail_ = false;
    return 0;
}

int Engine::process_batch(const int32_t* ids, const int32_t* id_lens, int B, int id_stride, const vits_process_opts& o, vits_batch_result* out,
                          std::string& err) {
    if (pending()) {
        err = "batches in flight: call vits_model_wait for every submitted batch first";
        return -1;
    }
    // The drop-in callers' share of the pipeline: a large batch is split in two inside the call — stage one of the
    // second part runs on the front-end stream under the flow / vocoder of the first, exactly as two vits_model_submit_batch calls would.
    // Every kernel is batch-invariant, so PCM, lengths and frames are those of the unsplit call bit for bit (GPU test, fuzz_identity.py).
    const int smin = knobs.split_min_batch;
    if (smin > 0 && B >= smin && B >= 2 && front_ && !prof.on && !knobs.no_pipeline && o.noise_kind == VITS_NOISE_COUNTER && !o.collect_taps && !o.on_chunk &&
        !o.frames_only && !o.async && o.vocoder_chunk_frames <= 0)
        return process_split(ids, id_lens, B, id_stride, o, out, err);
    a1_slot_ = 0;
    return process_impl(ids, id_lens, B, id_stride, o, out, err, nullptr);
}

int Engine::process_split(const int32_t* ids, const int32_t* id_lens, int B, int id_stride, const vits_process_opts& o, vits_batch_result* out,
                          std::string& err) {
    // first part: knobs.split_first_pct per cent of the utterances (its stage one is the only exposed one)
    const int B1 = std::min(B - 1, std::max(1, (int)(((int64_t)B * knobs.split_first_pct + 50) / 100)));
    const int nb[2] = {B1, B - B1}, b0[2] = {0, B1};
    std::vector<int32_t> offs(B);
    for (int b = 0; b < B; ++b) offs[b] = o.noise_seed_offsets ? o.noise_seed_offsets[b] : b;  // (a part keeps its utterances' global noise streams)
    int queued = 0;
    for (int h = 0; h < 2; ++h) {
        vits_process_opts oh = o;
        oh.noise_seed_offsets = offs.data() + b0[h];
        if (o.out_device) oh.out_device = (float*)o.out_device + (int64_t)b0[h] * o.out_device_stride;
        if (submit_batch(ids + (size_t)b0[h] * id_stride, id_lens ? id_lens + b0[h] : nullptr, nb[h], id_stride, oh, err)) break;
        ++queued;
    }
    if (queued < 2) {
        std::string ignored;
        while (pending()) wait_batch(nullptr, ignored);
        return -1;
    }
    // hand over: the parts' results side by side, rows at the stride of the longer part
    Pending* part[2] = {&pend_[wait_seq_ & 1], &pend_[(wait_seq_ + 1) & 1]};
    int rc = 0;
    for (int h = 0; h < 2 && !rc; ++h)
        if (hipEventSynchronize(part[h]->done) != hipSuccess) {
            err = "hipEventSynchronize failed";
            rc = -1;
        }
    // (the two slots are released whatever happens below: an allocation failure must not leave the handle with batches in flight)
    struct Release {
        Engine& e;
        Pending** part;
        ~Release() {
            for (int h = 0; h < 2; ++h) {
                part[h]->active = false;
                ++e.wait_seq_;
            }
        }
    } release{*this, part};
    if (!rc && out) {
        const size_t stride = std::max(part[0]->stride, part[1]->stride);
        out->batch = (size_t)B;
        out->stride = stride;
        out->lengths = new int64_t[B];
        out->frames = new int64_t[B];
        out->data = nullptr;
        if (!o.skip_host_copy) out->data = new float[(size_t)B * stride];
        for (int h = 0; h < 2; ++h) {
            const Pending& p = *part[h];
            std::copy(p.lengths.begin(), p.lengths.end(), out->lengths + b0[h]);
            std::copy(p.frames.begin(), p.frames.end(), out->frames + b0[h]);
            if (out->data)
                for (int b = 0; b < nb[h]; ++b) {
                    float* dst = out->data + (size_t)(b0[h] + b) * stride;
                    std::memcpy(dst, p.host + (size_t)b * p.stride, sizeof(float) * p.stride);
                    // (columns past a row's own length are unspecified in the unsplit result too; keep them defined)
                    if (p.stride < stride) std::memset(dst + p.stride, 0, sizeof(float) * (stride - p.stride));
                }
        }
    }
    return rc;
}

// ---- pipelined batches (vits.h: vits_model_submit_batch / vits_model_wait) --------------------------------------------------
// submit(n): stage one of batch n on the front-end stream into stage-one arena n & 1 — beside the flow / vocoder of batch n - 1,
// which is running on the main stream —, the frame counts through pinned memory behind an event (the host waits for THAT event,
// not for the device), stage two behind batch n - 1 on the main stream. Batch n - 2, the previous user of the arena, was waited
// for before this submit was accepted, so nothing on the device still reads it.
int Engine::submit_batch(const int32_t* ids, const int32_t* id_lens, int B, int id_stride, const vits_process_opts& o, std::string& err) {
    if (pending() >= 2) {
        err = "two batches in flight already: call vits_model_wait first";
        return -1;
    }
    if (o.noise_kind != VITS_NOISE_COUNTER || o.collect_taps || o.on_chunk || o.frames_only || o.async) {
        err = "vits_model_submit_batch: VITS_NOISE_COUNTER only, and no collect_taps / on_chunk / frames_only / async";
        return -1;
    }
    Pending& p = pend_[submit_seq_ & 1];
    if (!p.done) {
        HIP_OK(hipEventCreateWithFlags(&p.s1_done, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    }
    a1_slot_ = (int)(submit_seq_ & 1);
    const int rc = process_impl(ids, id_lens, B, id_stride, o, nullptr, err, &p);
    a1_slot_ = 0;
    if (rc) {
        // (whatever was queued is harmless: it only touches this slot's arenas; leave the device idle before the slot is reused)
        if (front_) hipStreamSynchronize(front_);
        hipStreamSynchronize(stream);
        return rc;
    }
    p.active = true;
    ++submit_seq_;
    return 0;
}

int Engine::wait_batch(vits_batch_result* out, std::string& err) {
    if (!pending()) {
        err = "vits_model_wait: nothing was submitted";
        return -1;
    }
    Pending& p = pend_[wait_seq_ & 1];
    HIP_OK(hipEventSynchronize(p.done));
    if (out) {
        // (the slot is released only once the result has been handed over: an allocation failure below leaves the batch waitable)
        out->batch = (size_t)p.B;
        out->stride = p.stride;
        out->lengths = new int64_t[p.B];
        out->frames = new int64_t[p.B];
        std::copy(p.lengths.begin(), p.lengths.end(), out->lengths);
        std::copy(p.frames.begin(), p.frames.end(), out->frames);
        out->data = nullptr;
        if (p.host_copy) {
            out->data = new float[(size_t)p.B * p.stride];
            std::memcpy(out->data, p.host, sizeof(float) * (size_t)p.B * p.stride);
        }
    }
    p.active = false;
    ++wait_seq_;
    return 0;
}

namespace {
// the stream member of the engine is "the stream being queued on": stage one of a pipelined batch swaps the front-end stream in
struct StreamSwap {
    hipStream_t& slot;
    hipStream_t saved;
    StreamSwap(hipStream_t& s, hipStream_t to) : slot(s), saved(s) { slot = to; }
    void restore() { slot = saved; }
    ~StreamSwap() { slot = saved; }
};
}  // namespace

int Engine::process_impl(const int32_t* ids, const int32_t* id_lens, int B, int id_stride, const vits_process_opts& o, vits_batch_result* out,
                         std::string& err, Pending* pend) {
    if (B <= 0 || id_stride <= 0) {
        err = "empty batch";
        return -1;
    }
    if (o.on_chunk && o.skip_host_copy) {  // (the only misuse of the sink; checked before any work is queued)
        err = "on_chunk needs a host copy (skip_host_copy = 0)";
        return -1;
    }
    Call c(o, err, ids, B, id_stride);
    // An error path between the speculative draw and its commit must not leave the reference stream locked — nor at a timing-dependent
    // position: finish(0) rewinds to the state behind the [T, 2] tensor, where the sequential path this replaces leaves the stream when a call
    // fails before prior sampling (finish(drawn()) committed however many values the helper happened to have drawn).
    struct AheadGuard {
        RefNoiseAhead& a;
        ~AheadGuard() {
            if (a.active()) a.finish(0);
        }
    } ahead_guard{ref_ahead_};
    c.md = o.mode == VITS_MODE_DEFAULT ? mode : o.mode;
    c.refmode = c.md == VITS_MODE_REFERENCE;
    c.tlen.resize(B);
    for (int b = 0; b < B; ++b) {
        c.tlen[b] = id_lens ? id_lens[b] : id_stride;
        if (c.tlen[b] <= 0 || c.tlen[b] > id_stride) {
            err = "bad id length";
            return -1;
        }
        c.Tmax = std::max(c.Tmax, c.tlen[b]);
        c.sum_t += c.tlen[b];
        for (int t = 0; t < c.tlen[b]; ++t) {
            const int id = ids[(size_t)b * id_stride + t];
            if (id < 0 || id >= hp.vocab_size) {
                err = "token id out of range";
                return -1;
            }
        }
    }
    if (!lat16_ready_ && knobs.lat16_lazy_tokens > 0 && (int64_t)B * c.Tmax <= knobs.lat16_lazy_tokens && ensure_lat16(err)) return -1;
    if (c.Tmax > 2048) {
        err = "more than 2048 ids per utterance is not supported";
        return -1;
    }
    const bool want_async = o.async && o.skip_host_copy && o.fixed_duration > 0 && !o.collect_taps;
    c.ts = round_up(c.Tmax, 32);
    const int n_up = c.n_up = (int)ups_.size();
    clear_taps();
    tap_batch_ = B;

    // ---- stage one: text encoder + duration predictor -------------------------------------------------------
    // Under VITS_ARITH_SCOPE_FLOW_VOCODER (default) stage one is exact fp32 in every arithmetic mode: the durations — the path's
    // only integer output, ceil() of a float (vits.cpp:996-1001) — are then bit-identical to the fp32 path's.
    arith_now_ = arith_scope == VITS_ARITH_SCOPE_ALL_CONVS ? arith_kernels() : VITS_ARITH_F32;
    // pipelined batch: stage one goes to the front-end stream (the per-kernel profiler needs kernels that do not overlap: then,
    // and under VITS_NO_PIPELINE, everything stays on the main stream and a pipelined batch is merely a deferred result)
    const bool overlap = pend && front_ && !prof.on;
    StreamSwap s1_stream(stream, overlap ? front_ : stream);
    hipStream_t const main_stream = s1_stream.saved;
    if (overlap && async_tail_) {
        // an earlier opts.async call returned without synchronising: its flow / vocoder may still be reading stage-one arena 0 on the
        // main stream, and nothing else orders the front-end stream behind it: stage one of this batch waits for that tail
        if (!ev_async_) HIP_OK(hipEventCreateWithFlags(&ev_async_, hipEventDisableTiming));
        HIP_OK(hipEventRecord(ev_async_, main_stream));
        HIP_OK(hipStreamWaitEvent(front_, ev_async_, 0));
    }
    if (B == 1 && o.noise_kind == VITS_NOISE_REFERENCE && !o.frames_only && knobs.ref_ahead_frames_per_id > 0) {
        // The reference's own call (vits_model_process: one utterance, libstdc++ noise). Its two noise tensors are drawn on the host: [T, 2] at
        // vits.cpp:948 and [L, 192] at :1059 — ~1.25 ms for 128 ids, and L is known only behind stage one. A helper thread draws the stream from NOW on,
        // while this thread queues stage one and the device runs it (RefNoiseAhead, engine_support.cpp); run_prior_sampling names the size.
        const size_t want = (size_t)hp.flow_size * ((size_t)knobs.ref_ahead_frames_per_id * c.Tmax + 64);
        if (ref_noise_cap_ < want) {
            if (ref_noise_pinned_) hipHostFree(ref_noise_pinned_);
            ref_noise_pinned_ = nullptr;
            ref_noise_cap_ = 0;
            HIP_OK(hipHostMalloc((void**)&ref_noise_pinned_, want * sizeof(float), hipHostMallocDefault));
            ref_noise_cap_ = want;
        }
        ref_ahead_.start((size_t)2 * c.tlen[0], ref_noise_pinned_, ref_noise_cap_);
        c.ref_ahead = &ref_ahead_;
    }
    if (layout_stage_one(c)) return -1;
    if (ggml_tables == 1) {
        // emulated-ggml mode: stage one in the exact order shared with the oracle (engine_stage1_exact.cpp)
        if (run_stage_one_exact(c)) return -1;
    } else {
        if (run_text_encoder(c)) return -1;
        if (run_duration_predictor(c)) return -1;
    }
    arith_now_ = arith_kernels();

    // ---- the one data-dependent shape (vits.cpp:1133): frames per utterance ---------------------------------
    c.rx.phase("vits.frame_count_sync");
    std::vector<int>& frames = c.frames;
    frames.resize(B);
    if (o.fixed_duration > 0) {
        for (int b = 0; b < B; ++b) frames[b] = std::max(1, o.fixed_duration * c.tlen[b]);
    } else if (pend) {
        // through pinned memory behind an event on the stage-one stream: the host waits for this batch's stage one only, while the
        // previous batch's vocoder keeps the device busy
        if (pend->frames_cap < (size_t)B) {
            if (pend->frames_pinned) hipHostFree(pend->frames_pinned);
            pend->frames_pinned = nullptr;
            pend->frames_cap = 0;
            HIP_OK(hipHostMalloc((void**)&pend->frames_pinned, sizeof(int) * (size_t)(B + 64), hipHostMallocDefault));
            pend->frames_cap = (size_t)B + 64;
        }
        HIP_OK(hipMemcpyAsync(pend->frames_pinned, c.s1.frames, sizeof(int) * B, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipEventRecord(pend->s1_done, stream));
        prof.fence();
        HIP_OK(hipEventSynchronize(pend->s1_done));
        std::copy(pend->frames_pinned, pend->frames_pinned + B, frames.begin());
    } else {
        // (pinned destination, owned by the engine: into the caller-side std::vector the copy went through the runtime's staging buffer, + 15 us per batch-1 call)
        if (frames_host_cap_ < (size_t)B) {
            if (frames_host_) hipHostFree(frames_host_);
            frames_host_ = nullptr;
            frames_host_cap_ = 0;
            HIP_OK(hipHostMalloc((void**)&frames_host_, sizeof(int) * ((size_t)B + 64), hipHostMallocDefault));
            frames_host_cap_ = (size_t)B + 64;
        }
        HIP_OK(hipMemcpyAsync(frames_host_, c.s1.frames, sizeof(int) * B, hipMemcpyDeviceToHost, stream));
        prof.fence();
        HIP_OK(hipStreamSynchronize(stream));
        prof.fence();
        std::copy(frames_host_, frames_host_ + B, frames.begin());
    }
    if (overlap) {
        // stage two runs on the main stream, behind this batch's stage one (pinned durations: no host read ordered them yet)
        if (o.fixed_duration > 0) HIP_OK(hipEventRecord(pend->s1_done, stream));
        HIP_OK(hipStreamWaitEvent(main_stream, pend->s1_done, 0));
    }
    s1_stream.restore();
    for (int b = 0; b < B; ++b) {
        c.Lmax = std::max(c.Lmax, frames[b]);
        c.sum_frames += frames[b];
    }
    const int Lmax = c.Lmax;
    c.slen.assign(n_up + 1, std::vector<int>(B));
    c.smax.assign(n_up + 1, 0);
    for (int i = 0; i <= n_up; ++i)
        for (int b = 0; b < B; ++b) {
            c.slen[i][b] = frames[b] * c.smul[i] + c.sadd[i];
            c.smax[i] = std::max(c.smax[i], c.slen[i][b]);
        }
    if (o.collect_taps) {
        TensorRef d;
        d.p = c.s1.dur;
        d.cs = id_stride;
        d.bs = id_stride;
        snapshot("durations", d, 1, c.Tmax, B, c.tlen);
    }
    if (o.frames_only) {
        // dispatcher query: predicted frames / samples per utterance, no audio (buffer sizing, shard balancing by frames)
        if (out) {
            out->batch = (size_t)B;
            out->stride = (size_t)c.smax[n_up];
            out->lengths = new int64_t[B];
            out->frames = new int64_t[B];
            out->data = nullptr;
            for (int b = 0; b < B; ++b) {
                out->lengths[b] = c.slen[n_up][b];
                out->frames[b] = frames[b];
            }
        }
        HIP_OK(hipStreamSynchronize(stream));
        prof.fence();
        return 0;
    }

    // ---- vocoder windows (long-form / streaming, vits.h vocoder_chunk_frames) ------------------------------------
    // Window w owns frames [f0, f1) and computes frames [lo, hi) = the owned range widened by the vocoder's receptive
    // field (halo_frames_): every sample it emits sees exactly the inputs it sees in a whole-utterance run, in the same
    // order of accumulation, so the PCM is bit-identical while the activations are bounded by the window.
    std::vector<Call::Win>& wins = c.wins;
    {
        const int W = (o.vocoder_chunk_frames > 0 && !o.collect_taps && o.vocoder_chunk_frames < Lmax) ? o.vocoder_chunk_frames : 0;
        if (!W) wins.push_back({0, Lmax, 0, Lmax});
        else
            for (int f0 = 0; f0 < Lmax; f0 += W) {
                const int f1 = std::min(Lmax, f0 + W);
                wins.push_back({f0, f1, std::max(0, f0 - halo_frames_), std::min(Lmax, f1 + halo_frames_)});
            }
    }
    const bool windowed = c.windowed = wins.size() > 1;
    // (a run that is not windowed — no chunking asked for, a window at least as long as the longest utterance, which the
    // caller cannot know in advance, or collect_taps — streams as ONE window: the sink gets each utterance in a single call)
    for (const Call::Win& w : wins) c.Lw_max = std::max(c.Lw_max, w.hi - w.lo);
    const int M = c.M = c.smul[n_up];  // samples per frame

    if (layout_stage_two(c)) return -1;
    if (run_prior_sampling(c)) return -1;
    if (run_flow(c)) return -1;

    // ---- HiFiGAN (vits.cpp:583-644), window by window ---------------------------------------------------------------
    c.rx.phase("vits.hifigan");
    const std::vector<int>& smax = c.smax;
    const std::vector<int>& smul = c.smul;
    const std::vector<int>& sadd = c.sadd;
    c.wave_dst = c.s2.wave;
    c.wave_stride = c.S_stride;
    if (o.out_device) {
        if (o.out_device_stride < smax[n_up]) {
            err = "out_device_stride is smaller than the longest utterance";
            return -1;
        }
        c.wave_dst = (float*)o.out_device;
        c.wave_stride = o.out_device_stride;
    }
    float* const wave_dst = c.wave_dst;
    const int64_t wave_stride = c.wave_stride;
    std::vector<hipEvent_t> chunk_ev;
    struct EvGuard {
        std::vector<hipEvent_t>& v;
        ~EvGuard() {
            for (hipEvent_t e : v) hipEventDestroy(e);
        }
    } chunk_ev_guard{chunk_ev};
    float* host_pcm = nullptr;  // pinned [B][smax] staging of the streamed PCM
    const size_t out_stride = (size_t)smax[n_up];
    if (windowed) {
        // per window and utterance: frames inside the window -> stage lengths (0 everywhere when the utterance has none),
        // and the end of the sample range this window emits for it (window-local index)
        std::vector<int> wl(wins.size() * (size_t)(n_up + 2) * B, 0);
        for (size_t w = 0; w < wins.size(); ++w)
            for (int b = 0; b < B; ++b) {
                const Call::Win& wn = wins[w];
                const int lf = std::min(frames[b], wn.hi) - wn.lo;
                int* row = wl.data() + w * (size_t)(n_up + 2) * B;
                if (lf <= 0) continue;
                for (int i = 0; i <= n_up; ++i) row[(size_t)i * B + b] = lf * smul[i] + sadd[i];
                if (frames[b] > wn.f0)  // owns frames here; the window holding the utterance's end also emits its tail (Q1)
                    row[(size_t)(n_up + 1) * B + b] = frames[b] <= wn.f1 ? (frames[b] - wn.lo) * M + sadd[n_up] : (wn.f1 - wn.lo) * M;
            }
        if (pend) {
            // a pipelined batch: the table travels through pinned memory owned by the slot (it outlives this call), so that the submit does not
            // drain the main stream — the previous batch's vocoder is running there (the synchronise below serialised windowed submits)
            if (pend->win_cap < wl.size()) {
                if (pend->win_pinned) hipHostFree(pend->win_pinned);
                pend->win_pinned = nullptr;
                pend->win_cap = 0;
                HIP_OK(hipHostMalloc((void**)&pend->win_pinned, sizeof(int) * (wl.size() + wl.size() / 4 + 64), hipHostMallocDefault));
                pend->win_cap = wl.size() + wl.size() / 4 + 64;
            }
            std::memcpy(pend->win_pinned, wl.data(), sizeof(int) * wl.size());
            HIP_OK(hipMemcpyAsync(c.s2.win_lens, pend->win_pinned, sizeof(int) * wl.size(), hipMemcpyHostToDevice, stream));
            prof.fence();
        } else {
            HIP_OK(hipMemcpyAsync(c.s2.win_lens, wl.data(), sizeof(int) * wl.size(), hipMemcpyHostToDevice, stream));
            prof.fence();
            HIP_OK(hipStreamSynchronize(stream));  // wl goes out of scope
        }
    }
    if (o.on_chunk) {
        const size_t need = (size_t)B * out_stride * sizeof(float);
        if (need > pinned_cap_) {
            if (pinned_) hipHostFree(pinned_);
            pinned_ = nullptr;
            pinned_cap_ = 0;
            HIP_OK(hipHostMalloc((void**)&pinned_, need, hipHostMallocDefault));
            pinned_cap_ = need;
        }
        host_pcm = (float*)pinned_;
    }
    // hands the finished window `w` to the caller's sink (blocks until its PCM is on the host)
    auto deliver = [&](size_t w) -> int {
        const Call::Win& wn = wins[w];
        if (hipEventSynchronize(chunk_ev[w]) != hipSuccess) return -1;
        for (int b = 0; b < B; ++b) {
            if (frames[b] <= wn.f0) continue;
            const size_t off = (size_t)wn.f0 * M;
            const size_t end = frames[b] <= wn.f1 ? (size_t)c.slen[n_up][b] : (size_t)wn.f1 * M;
            if (o.on_chunk(o.on_chunk_user, b, off, host_pcm + (size_t)b * out_stride + off, end - off)) return 1;
        }
        return 0;
    };
    for (size_t wi = 0; wi < wins.size(); ++wi) {
        WinCtx w;
        w.wi = wi;
        w.wn = wins[wi];
        const Call::Win& wn = w.wn;
        w.Lw = wn.hi - wn.lo;
        w.smax.resize(n_up + 1);  // (window-local maxima: everything inside the window functions is window-local)
        for (int i = 0; i <= n_up && i < 8; ++i) {
            w.d_len[i] = windowed ? c.s2.win_lens + (wi * (size_t)(n_up + 2) + i) * B : c.d_len_full[i];
            w.smax[i] = w.Lw * smul[i] + sadd[i];
        }
        w.ssum.assign(n_up + 1, 0);
        for (int b = 0; b < B; ++b) {
            const int lf = std::min(frames[b], wn.hi) - wn.lo;
            if (lf <= 0) continue;
            for (int i = 0; i <= n_up; ++i) w.ssum[i] += (int64_t)lf * smul[i] + sadd[i];
        }
        w.emit_hi = windowed ? c.s2.win_lens + (wi * (size_t)(n_up + 2) + n_up + 1) * B : nullptr;
        w.emit_lo = (wn.f0 - wn.lo) * M;
        w.zwin = make_ref(c.s2.zp, hp.flow_size, c.ls);
        w.zwin.p += wn.lo;
        w.pre.p = c.s2.pre;
        w.pre.bs = c.S_stride;
        w.pre.cs = c.S_stride;
        w.wv.p = wave_dst + (int64_t)wn.lo * M;  // window-local sample 0 is global sample lo * M
        w.wv.bs = wave_stride;
        w.wv.cs = (int)wave_stride;
        w.final_slope = c.refmode ? hp.lrelu : 0.01f;  // Q2 (vits.cpp:638)
        if (c.fast16 ? run_vocoder_window16(c, w) : run_vocoder_window32(c, w)) return -1;
        if (o.on_chunk) {
            // ship this window's samples to the host behind the kernels, then serve the PREVIOUS window's callbacks while
            // the device works on this one
            // (+ the no-crop tail of utterances that end inside this window, Q1; the same columns of longer utterances are not
            // final yet and travel again with the next window)
            const size_t g0 = (size_t)wn.f0 * M, g1 = std::min(out_stride, (size_t)wn.f1 * M + (size_t)sadd[n_up]);
            HIP_OK(hipMemcpy2DAsync(host_pcm + g0, out_stride * 4, wave_dst + g0, (size_t)wave_stride * 4, (g1 - g0) * 4, (size_t)B, hipMemcpyDeviceToHost, stream));
            prof.fence();
            hipEvent_t ev;
            HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            chunk_ev.push_back(ev);
            HIP_OK(hipEventRecord(ev, stream));
            if (wi > 0)
                if (int rc = deliver(wi - 1)) {
                    hipStreamSynchronize(stream);
                    err = rc > 0 ? "aborted by the on_chunk callback" : "hipEventSynchronize failed";
                    return -1;
                }
        }
        if (o.collect_taps) {
            snapshot("pre_tanh", w.pre, 1, w.smax[n_up], B, c.slen[n_up]);
            snapshot("waveform", w.wv, 1, w.smax[n_up], B, c.slen[n_up]);
        }
    }
    if (o.on_chunk)
        if (int rc = deliver(wins.size() - 1)) {
            hipStreamSynchronize(stream);
            err = rc > 0 ? "aborted by the on_chunk callback" : "hipEventSynchronize failed";
            return -1;
        }

    // ---- results ------------------------------------------------------------------------------------------------
    c.rx.phase("vits.results");
    if (pend) {
        // a pipelined batch: everything the host will hand out is known now; the PCM (if a host copy was asked for) goes to the
        // slot's pinned staging behind the kernels, and `done` marks the end of the batch on the device
        pend->B = B;
        pend->stride = (size_t)smax[n_up];
        pend->lengths.resize(B);
        pend->frames.resize(B);
        for (int b = 0; b < B; ++b) {
            pend->lengths[b] = c.slen[n_up][b];
            pend->frames[b] = frames[b];
        }
        pend->host_copy = !o.skip_host_copy;
        if (pend->host_copy) {
            const size_t need = sizeof(float) * (size_t)B * pend->stride;
            if (need > pend->host_cap) {
                if (pend->host) hipHostFree(pend->host);
                pend->host = nullptr;
                pend->host_cap = 0;
                HIP_OK(hipHostMalloc((void**)&pend->host, need + need / 8, hipHostMallocDefault));
                pend->host_cap = need + need / 8;
            }
            HIP_OK(hipMemcpy2DAsync(pend->host, pend->stride * 4, wave_dst, (size_t)wave_stride * 4, pend->stride * 4, (size_t)B, hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipEventRecord(pend->done, stream));
        prof.fence();
        return 0;
    }
    if (out) {
        out->batch = (size_t)B;
        out->stride = (size_t)smax[n_up];
        out->lengths = new int64_t[B];
        out->frames = new int64_t[B];
        for (int b = 0; b < B; ++b) {
            out->lengths[b] = c.slen[n_up][b];
            out->frames[b] = frames[b];
        }
        out->data = nullptr;
        if (!o.skip_host_copy) {
            out->data = new float[(size_t)B * out->stride];
            if (o.on_chunk) std::memcpy(out->data, host_pcm, sizeof(float) * (size_t)B * out->stride);  // already streamed to the host
            else
                HIP_OK(hipMemcpy2DAsync(out->data, out->stride * 4, wave_dst, (size_t)wave_stride * 4, out->stride * 4, (size_t)B, hipMemcpyDeviceToHost, stream));
            prof.fence();
        }
    }
    async_tail_ = want_async;
    if (!want_async) {
        HIP_OK(hipStreamSynchronize(stream));
        prof.fence();
        // (event timings are read back lazily — vits_prof_report / vits_prof_reset — not here: ~1 ms of host work per call
        // with the device idle would otherwise sit inside the caller's timed region; cap the backlog for long runs)
        if (prof.on && prof.recs.size() > 50000) prof.collect();
    }
    return 0;
}

}  // namespace vits
