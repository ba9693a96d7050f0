// engine_load.cpp — weights: shape-checked upload, MFMA-fragment packing (fp32 at load, 16-bit on vits_model_set_arith), taps.
#include "engine_internal.h"
#include "../../include/vits_exact_math.h"

namespace vits {

// ---- load -------------------------------------------------------------------------------------------------
Engine::~Engine() {
    if (stream) hipStreamSynchronize(stream);
    if (front_) hipStreamSynchronize(front_);
    for (hipStream_t s : side_)
        if (s) hipStreamSynchronize(s);
    clear_taps();
    for (Pending& p : pend_) {
        if (p.host) hipHostFree(p.host);
        if (p.frames_pinned) hipHostFree(p.frames_pinned);
        if (p.win_pinned) hipHostFree(p.win_pinned);
        if (p.s1_done) hipEventDestroy(p.s1_done);
        if (p.done) hipEventDestroy(p.done);
    }
    if (front_) hipStreamDestroy(front_);
    if (!dry_run_) {
        for (void* p : owned_) hipFree(p);
        for (PackSrc& ps : packs_)
            if (ps.pc->wp16) hipFree(ps.pc->wp16);
    }
    if (pinned_) hipHostFree(pinned_);
    if (frames_host_) hipHostFree(frames_host_);
    for (HStage& hs : hstage_) {
        if (hs.p) hipHostFree(hs.p);
        if (hs.ev) hipEventDestroy(hs.ev);
    }
    if (ev_async_) hipEventDestroy(ev_async_);
    if (ref_noise_pinned_) hipHostFree(ref_noise_pinned_);
    if (dur_noise_pinned_) hipHostFree(dur_noise_pinned_);
    if (dur_noise_ev_) hipEventDestroy(dur_noise_ev_);
    if (ev_fork_) hipEventDestroy(ev_fork_);
    for (hipEvent_t e : ev_done_)
        if (e) hipEventDestroy(e);
    for (hipStream_t s : side_)
        if (s) hipStreamDestroy(s);
    if (stream) hipStreamDestroy(stream);
}

// ggml_init builds table_gelu_f16 / table_exp_f16 on the host by evaluating ggml_gelu_f32 / expf on every fp16 value (upstream ggml.c of the
// reference's era; GGML_GELU_FP16 is on by default) — the same here, with this host's C library, then the 2 x 128 KB go to the device.
int Engine::set_ggml_tables(int mode, std::string& err) {
    if (mode < 0 || mode > 2) {
        err = "vits_model_set_ggml_tables: 0 (off), 1 (tables, stage one in the exact shared order) or 2 (tables inside the throughput kernels)";
        return -1;
    }
    if (mode && !ggml_tab_dev_) {
        // built on the HOST with the C library's tanhf / expf, as ggml_init does — by the function the oracle builds its copy with
        std::vector<uint16_t> tab(2 * 65536);
        vx_build_ggml_tables(tab.data(), tab.data() + 65536);
        uint16_t* d = nullptr;
        if (hipMalloc((void**)&d, tab.size() * sizeof(uint16_t)) != hipSuccess || hipMemcpy(d, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) {
            if (d) hipFree(d);
            err = "could not upload the ggml lookup tables";
            return -1;
        }
        ggml_tab_dev_ = d;
        owned_.push_back(d);
    }
    if (mode == 1 && exact_w_.empty()) {
        // the stage-one tensors in torch layout, fp32 on the device (uploaded once per handle)
        for (const TensorEntry& t : exact_src_) {
            float* d = upload(t.to_f32());
            if (!d) {
                err = "hipMalloc failed for the exact-order copy of " + t.name;
                return -1;
            }
            ExactTensor e;
            e.d = d;
            e.rank = (int)t.rank;
            for (int k = 0; k < 4; ++k) e.ne[k] = t.ne[k];
            exact_w_[t.name] = e;
        }
        if (hipDeviceSynchronize() != hipSuccess) {
            err = "device error while uploading the exact-order weights";
            return -1;
        }
    }
    ggml_tables = mode;
    ggml_tabs_ = GgmlTables();
    if (mode) {
        ggml_tabs_.gelu = ggml_tab_dev_;
        ggml_tabs_.exp = ggml_tab_dev_ + 65536;
    }
    return 0;
}

float* Engine::upload(const std::vector<float>& v) {
    if (dry_run_) return reinterpret_cast<float*>(16);  // validation only (vits_model_file_validate): nothing is allocated
    float* d = nullptr;
    if (hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * sizeof(float)) != hipSuccess) return nullptr;
    hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
    owned_.push_back(d);
    weight_bytes += (int64_t)v.size() * 4;
    return d;
}

// Shape check against the hyper-parameters (file order: fastest dimension first; missing trailing dimensions count as 1;
// -1 = any). The reference trusts the file (ggml asserts or reads out of bounds, vits_model_data.cpp:56-89); here a tensor
// whose shape does not match what the kernels will index with is a load error, never a device out-of-bounds access.
static bool shape_is(const TensorEntry& t, std::initializer_list<int64_t> want) {
    if (want.size() < t.rank) {
        for (uint32_t j = (uint32_t)want.size(); j < t.rank; ++j)
            if (t.ne[j] != 1) return false;
    }
    size_t j = 0;
    for (int64_t w : want) {
        const int64_t have = j < t.rank ? t.ne[j] : 1;
        if (w >= 0 && have != w) return false;
        ++j;
    }
    return true;
}
static std::string shape_str(const TensorEntry& t) {
    std::string o = "[";
    for (uint32_t j = 0; j < t.rank; ++j) o += (j ? "," : "") + std::to_string(t.ne[j]);
    return o + "]";
}
static std::string shape_str(std::initializer_list<int64_t> want) {
    std::string o = "[";
    size_t j = 0;
    for (int64_t w : want) o += (j++ ? "," : "") + (w < 0 ? std::string("*") : std::to_string(w));
    return o + "]";
}

float* Engine::upload_tensor(const ModelFile& f, const std::string& name, std::string& err, std::initializer_list<int64_t> want) {
    const TensorEntry* t = f.find(name);
    if (!t) {
        err = "[ERROR] tensor not found: " + name;  // message of the reference, vits_model_data.cpp:144
        return nullptr;
    }
    if (!shape_is(*t, want)) {
        err = "tensor '" + name + "' has shape " + shape_str(*t) + ", the hyper-parameters need " + shape_str(want);
        return nullptr;
    }
    float* d = upload(t->to_f32());
    if (!d) err = "hipMalloc failed for " + name;
    return d;
}

static bool get_conv(const ModelFile& f, const std::string& wname, std::vector<float>& w, int& cout, int& cin, int& k, uint32_t& dtype, std::string& err) {
    const TensorEntry* t = f.find(wname);
    if (!t) {
        err = "[ERROR] tensor not found: " + wname;
        return false;
    }
    w = t->to_f32();
    dtype = t->dtype;
    if (t->rank == 3) {  // file ne = [k, cin, cout] (reversed torch [cout][cin][k])
        k = (int)t->ne[0];
        cin = (int)t->ne[1];
        cout = (int)t->ne[2];
    } else if (t->rank == 2) {  // Linear [out][in]
        k = 1;
        cin = (int)t->ne[0];
        cout = (int)t->ne[1];
    } else {
        err = "unexpected rank for " + wname;
        return false;
    }
    return true;
}

// transform: 0 none | 1 reverse input channels | 2 negate | 3 negate + reverse output channels
// want = {cout, cin, k} the hyper-parameters imply (-1: taken from the file)
bool Engine::pack(const ModelFile& f, const std::string& wname, const std::string& bname, int epi, ConvShape want, PackedConv& out, std::string& err,
                  int ct_stride, int transform) {
    std::vector<float> w;
    int d0, d1, k;
    uint32_t dtype = DT_F32;
    if (!get_conv(f, wname, w, d0, d1, k, dtype, err)) return false;
    int cout = d0, cin = d1;
    if (epi == EPI_CONVT) {  // torch ConvTranspose1d weight [cin][cout][k]
        cin = d0;
        cout = d1;
    }
    if ((want.cout >= 0 && cout != want.cout) || (want.cin >= 0 && cin != want.cin) || (want.k >= 0 && k != want.k) || cout <= 0 || cin <= 0 || k <= 0 ||
        (epi == EPI_CONVT && (ct_stride <= 0 || k != 2 * ct_stride)) || (epi == EPI_GATE && (cout & 1))) {
        err = "tensor '" + wname + "' is a " + std::to_string(cout) + "x" + std::to_string(cin) + "x" + std::to_string(k) + " kernel (out x in x taps), the hyper-parameters need " +
              (want.cout < 0 ? std::string("*") : std::to_string(want.cout)) + "x" + (want.cin < 0 ? std::string("*") : std::to_string(want.cin)) + "x" +
              (want.k < 0 ? std::string("*") : std::to_string(want.k));
        return false;
    }
    std::vector<float> bias;
    if (!bname.empty()) {
        const TensorEntry* b = f.find(bname);
        if (!b) {
            err = "[ERROR] tensor not found: " + bname;
            return false;
        }
        if (b->count() != cout) {
            err = "tensor '" + bname + "' has " + std::to_string(b->count()) + " elements, expected " + std::to_string(cout);
            return false;
        }
        bias = b->to_f32();
    }
    if (transform == 1) {
        std::vector<float> w2(w.size());
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int j = 0; j < k; ++j) w2[((size_t)co * cin + ci) * k + j] = w[((size_t)co * cin + (cin - 1 - ci)) * k + j];
        w.swap(w2);
    } else if (transform == 2 || transform == 3) {
        std::vector<float> w2(w.size()), b2(bias.size());
        for (int co = 0; co < cout; ++co) {
            const int src = transform == 3 ? cout - 1 - co : co;
            for (int e = 0; e < cin * k; ++e) w2[(size_t)co * cin * k + e] = -w[(size_t)src * cin * k + e];
            if (!bias.empty()) b2[co] = -bias[src];
        }
        w.swap(w2);
        bias.swap(b2);
    }
    out.cin = cin;
    out.cout = cout;
    out.epi = epi;
    out.ct_stride = ct_stride;
    out.kt = epi == EPI_CONVT ? k / ct_stride : k;
    std::vector<float> packed = pack_conv_weights(w.data(), cout, cin, k, epi, ct_stride, &out.rows, &out.mtiles_used, &out.mtiles, &out.nchunks);
    if (!dry_run_) {
        // source of the 16-bit fragments set_arith packs on demand, kept in the file's own storage type (the flips / negations
        // above are exact in it): 2 bytes per parameter for the fp16 / bf16 files the exporter writes, not a second fp32 copy
        PackSrc ps{&out, {}, {}, dtype, cout, cin, k, epi, ct_stride};
        if (dtype == DT_F16 || dtype == DT_BF16) {
            ps.w16.resize(w.size());
            for (size_t e = 0; e < w.size(); ++e) ps.w16[e] = dtype == DT_F16 ? f32_to_f16(w[e]) : f32_to_bf16(w[e]);
        } else
            ps.w32 = w;
        packs_.push_back(std::move(ps));
    }
    out.wp = upload(packed);
    // the second copy of the weights for the latency kernels (batch 1 / short inputs: conv_lat16_kernel, stage1_lat.hip) is made by ensure_lat16() at the
    // first SMALL call (or at load under VITS_LAT16_EAGER=1): a handle that only ever serves large batches keeps ONE fp32 copy of its weights
    if (!dry_run_ && conv_lat16_candidate(epi, out.kt, cin)) lat16_lazy_.push_back({&out, packed.size()});
    out.bias = bias.empty() ? nullptr : upload(bias);
    out.bytes = (int64_t)packed.size() * 4;  // what ONE launch reads (a launch takes wp or wp_l16, never both; weight_bytes counts both copies)
    if (!out.wp) {
        err = "hipMalloc failed for " + wname;
        return false;
    }
    return true;
}

bool Engine::load_dds(const ModelFile& f, const std::string& base, DdsW& d, std::string& err) {
    const int H = hp.hidden;
    d.pw.resize(hp.dds_layers);  // (sized first: set_arith keeps pointers to the PackedConv entries)
    for (int i = 0; i < hp.dds_layers; ++i) {
        const std::string si = std::to_string(i);
        float* p;
        if (!(p = upload_tensor(f, base + "convs_dilated." + si + ".weight", err, {hp.dp_k, 1, H}))) return false;  // depthwise: torch [H][1][k]
        d.dw_w.push_back(p);
        if (!(p = upload_tensor(f, base + "convs_dilated." + si + ".bias", err, {H}))) return false;
        d.dw_b.push_back(p);
        if (!pack(f, base + "convs_pointwise." + si + ".weight", base + "convs_pointwise." + si + ".bias", EPI_STD, {H, H, 1}, d.pw[i], err)) return false;
        if (!(p = upload_tensor(f, base + "norms_1." + si + ".weight", err, {H}))) return false;
        d.n1_g.push_back(p);
        if (!(p = upload_tensor(f, base + "norms_1." + si + ".bias", err, {H}))) return false;
        d.n1_b.push_back(p);
        if (!(p = upload_tensor(f, base + "norms_2." + si + ".weight", err, {H}))) return false;
        d.n2_g.push_back(p);
        if (!(p = upload_tensor(f, base + "norms_2." + si + ".bias", err, {H}))) return false;
        d.n2_b.push_back(p);
    }
    return true;
}

bool Engine::load(const uint8_t* bytes, size_t size, std::string& err) {
    ModelFile f;
    if (!f.parse(bytes, size, err)) return false;
    if (!hp.load(f, err)) return false;
    tok.init(f);
    int ndev = 0;
    if (!dry_run_ && (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)) {
        err = "no HIP device available: this library has no CPU path";
        return false;
    }
    if (!dry_run_ && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
        err = "hipStreamCreate failed";
        return false;
    }
    knobs.read();
    KernelKnobsScope kernel_knobs_scope(&knobs.kernel);  // (load decides which fused kernels a stage can take)
    prof.attach = knobs.prof_attach;
    if (!dry_run_ && !knobs.no_pipeline) {
        // The front-end stream of pipelined batches (vits_model_submit_batch) is created HERE, right behind the main stream, not on first
        // use: HIP maps streams onto a small pool of hardware queues as they are created, and a front-end stream created late — after the
        // side streams and whatever the host application (torch) has opened — can land on the hardware queue of the main stream, where stage one
        // of batch i + 1 then queues BEHIND the vocoder of batch i instead of beside it (measured inside bench.py, where the model is loaded before
        // torch's first allocation: 14.2 ms per f16 batch against 13.45 with the stream created here; serial calls 14.65 either way).
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&front_, hipStreamNonBlocking, knobs.front_prio ? greatest : least) != hipSuccess) {
            front_ = nullptr;  // (no priorities on this device: a plain stream does the same job — the priority measured +-0)
            if (hipStreamCreateWithFlags(&front_, hipStreamNonBlocking) != hipSuccess) {
                err = "hipStreamCreate failed";
                return false;
            }
        }
    }
    if (knobs.rb_streams > 1 && !dry_run_) {
        bool ok = hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming) == hipSuccess;
        for (auto& s : side_) ok = ok && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
        for (auto& ev : ev_done_) ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            err = "hipStreamCreate failed";
            return false;
        }
    }
    const int H = hp.hidden, F = hp.flow_size;
    // structural limits of the kernels (what the shapes below are checked against)
    if (H <= 0 || hp.heads <= 0 || H % hp.heads != 0 || F <= 0 || (F & 1) || hp.window < 0 || hp.layers < 0 || hp.up_rates.size() != hp.up_k.size() ||
        hp.up_rates.size() > 6 || hp.rb_k.empty() || hp.rb_k.size() != hp.rb_d.size() || hp.dp_bins <= 0 || hp.dp_k <= 0 || hp.dds_layers < 0 ||
        hp.dp_flows < 1 || hp.n_flows < 0 || hp.wn_layers < 1) {
        err = "unsupported hyper-parameters";
        return false;
    }
    {
        const TensorEntry* e = f.find("text_encoder.embed_tokens.weight");
        if (!e || e->rank != 2 || e->ne[0] != H || e->ne[1] <= 0) {
            err = e ? "tensor 'text_encoder.embed_tokens.weight' must be [hidden, vocab]" : "[ERROR] tensor not found: text_encoder.embed_tokens.weight";
            return false;
        }
        hp.vocab_size = (int)e->ne[1];
    }
    if (!(emb_ = upload_tensor(f, "text_encoder.embed_tokens.weight", err, {H, hp.vocab_size}))) return false;
    const int hd = H / hp.heads, nrel = 2 * hp.window + 1;
    enc_.resize(hp.layers);
    for (int l = 0; l < hp.layers; ++l) {
        const std::string b = "text_encoder.encoder.layers." + std::to_string(l) + ".";
        EncoderLayerW& L = enc_[l];
        // fused Q|K|V projection: one GEMM with 3H output rows (vits.cpp:287-289 are three mul_mat + three adds)
        {
            std::vector<float> w((size_t)3 * H * H), bias((size_t)3 * H);
            const char* names[3] = {"q_proj", "k_proj", "v_proj"};
            for (int i = 0; i < 3; ++i) {
                const TensorEntry* tw = f.find(b + "attention." + names[i] + ".weight");
                const TensorEntry* tb = f.find(b + "attention." + names[i] + ".bias");
                if (!tw || !tb) {
                    err = "[ERROR] tensor not found: " + b + "attention." + names[i];
                    return false;
                }
                if (!shape_is(*tw, {H, H}) || !shape_is(*tb, {H})) {
                    err = "tensor '" + b + "attention." + names[i] + "' must be a [hidden, hidden] Linear with a [hidden] bias";
                    return false;
                }
                auto wv = tw->to_f32();
                auto bv = tb->to_f32();
                std::memcpy(w.data() + (size_t)i * H * H, wv.data(), sizeof(float) * H * H);
                std::memcpy(bias.data() + (size_t)i * H, bv.data(), sizeof(float) * H);
            }
            PackedConv& pc = L.qkv;
            pc.cin = H;
            pc.cout = 3 * H;
            pc.kt = 1;
            pc.epi = EPI_STD;
            auto packed = pack_conv_weights(w.data(), 3 * H, H, 1, EPI_STD, 0, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
            pc.wp = upload(packed);
            if (!dry_run_ && conv_lat16_candidate(EPI_STD, 1, H)) lat16_lazy_.push_back({&pc, packed.size()});
            pc.bias = upload(bias);
            pc.bytes = (int64_t)packed.size() * 4;
            if (!pc.wp || !pc.bias) {
                err = "hipMalloc failed for " + b + "attention";
                return false;
            }
        }
        if (!pack(f, b + "attention.out_proj.weight", b + "attention.out_proj.bias", EPI_STD, {H, H, 1}, L.out, err)) return false;
        if (!dry_run_) packs_.pop_back();  // a Linear (ggml_mul_mat on f32 x f32, vits.cpp:358), not a conv: no 16-bit operands in any mode
        if (!pack(f, b + "feed_forward.conv_1.weight", b + "feed_forward.conv_1.bias", EPI_STD, {hp.ffn_dim, H, hp.ffn_k}, L.ffn1, err)) return false;
        if (!pack(f, b + "feed_forward.conv_2.weight", b + "feed_forward.conv_2.bias", EPI_STD, {H, hp.ffn_dim, hp.ffn_k}, L.ffn2, err)) return false;
        if (!(L.rel_k = upload_tensor(f, b + "attention.emb_rel_k", err, {hd, nrel, 1}))) return false;  // shared by the heads (vits.cpp:323)
        if (!(L.rel_v = upload_tensor(f, b + "attention.emb_rel_v", err, {hd, nrel, 1}))) return false;
        if (!(L.ln1_g = upload_tensor(f, b + "layer_norm.weight", err, {H}))) return false;
        if (!(L.ln1_b = upload_tensor(f, b + "layer_norm.bias", err, {H}))) return false;
        if (!(L.ln2_g = upload_tensor(f, b + "final_layer_norm.weight", err, {H}))) return false;
        if (!(L.ln2_b = upload_tensor(f, b + "final_layer_norm.bias", err, {H}))) return false;
    }
    if (!pack(f, "text_encoder.project.weight", "text_encoder.project.bias", EPI_STD, {2 * F, H, 1}, enc_proj_, err)) return false;
    // duration predictor
    {
        const std::string dp = "duration_predictor.";
        if (!pack(f, dp + "conv_pre.weight", dp + "conv_pre.bias", EPI_STD, {H, H, 1}, dp_pre_, err)) return false;
        if (!pack(f, dp + "conv_proj.weight", dp + "conv_proj.bias", EPI_STD, {H, H, 1}, dp_proj_, err)) return false;
        if (!load_dds(f, dp + "conv_dds.", dp_dds_, err)) return false;
        if (!(dp_translate_ = upload_tensor(f, dp + "flows.0.translate", err, {1, 2}))) return false;
        if (!(dp_logscale_ = upload_tensor(f, dp + "flows.0.log_scale", err, {1, 2}))) return false;
        dp_flows_.resize(hp.dp_flows);
        for (int fl = 1; fl <= hp.dp_flows; ++fl) {
            if (fl == 1) continue;  // never evaluated (vits.cpp:954; HF "remove a useless vflow")
            const std::string b = dp + "flows." + std::to_string(fl) + ".";
            DpFlowW& W = dp_flows_[fl - 1];
            if (!(W.pre_w = upload_tensor(f, b + "conv_pre.weight", err, {1, 1, H}))) return false;  // Conv1d(1 -> H, 1) (vits.cpp:864)
            if (!(W.pre_b = upload_tensor(f, b + "conv_pre.bias", err, {H}))) return false;
            if (!load_dds(f, b + "conv_dds.", W.dds, err)) return false;
            if (!pack(f, b + "conv_proj.weight", b + "conv_proj.bias", EPI_STD, {3 * hp.dp_bins - 1, H, 1}, W.proj, err)) return false;
        }
    }
    // coupling flow: channel flips (vits.cpp:532) are folded into the weights. Layer i (processed i = n-1 .. 0) sees
    // (n - i) flips; with an odd count the logical first half lives in physical channels [F/2, F) reversed.
    flow_.resize(hp.n_flows);
    for (int i = 0; i < hp.n_flows; ++i) {
        const std::string b = "flow.flows." + std::to_string(i) + ".";
        const bool flipped = ((hp.n_flows - i) % 2) == 1;
        FlowLayerW& L = flow_[i];
        if (!pack(f, b + "conv_pre.weight", b + "conv_pre.bias", EPI_STD, {H, F / 2, 1}, L.pre, err, 0, flipped ? 1 : 0)) return false;
        if (!pack(f, b + "conv_post.weight", b + "conv_post.bias", EPI_STD, {F / 2, H, 1}, L.post, err, 0, flipped ? 3 : 2)) return false;  // x1 -= mean
        L.in_layers.resize(hp.wn_layers);
        L.res_skip.resize(hp.wn_layers);
        for (int l = 0; l < hp.wn_layers; ++l) {
            const std::string sl = std::to_string(l);
            if (!pack(f, b + "wavenet.in_layers." + sl + ".weight", b + "wavenet.in_layers." + sl + ".bias", EPI_GATE, {2 * H, H, hp.wn_k}, L.in_layers[l], err)) return false;
            if (!pack(f, b + "wavenet.res_skip_layers." + sl + ".weight", b + "wavenet.res_skip_layers." + sl + ".bias", EPI_STD,
                      {l + 1 < hp.wn_layers ? 2 * H : H, H, 1}, L.res_skip[l], err))
                return false;
        }
    }
    // HiFiGAN
    if (!pack(f, "decoder.conv_pre.weight", "decoder.conv_pre.bias", EPI_STD, {hp.up_init, F, -1}, dec_pre_, err)) return false;
    if (!(dec_pre_.kt & 1)) {
        err = "decoder.conv_pre needs an odd kernel size";
        return false;
    }
    ups_.resize(hp.up_rates.size());
    {
        int c = hp.up_init;
        for (size_t i = 0; i < hp.up_rates.size(); ++i) {
            UpStageW& U = ups_[i];
            U.stride = hp.up_rates[i];
            U.k = hp.up_k[i];
            if (U.stride <= 0 || U.k != 2 * U.stride || (c & 1)) {
                err = "unsupported upsampler (kernel size must be twice the stride)";
                return false;
            }
            const int cin_stage = c;
            c /= 2;
            U.channels = c;
            const std::string si = std::to_string(i);
            if (!pack(f, "decoder.upsampler." + si + ".weight", "decoder.upsampler." + si + ".bias", EPI_CONVT, {c, cin_stage, U.k}, U.up, err, U.stride)) return false;
            U.rbs.resize(hp.rb_k.size());
            for (size_t j = 0; j < hp.rb_k.size(); ++j) {
                ResBlockW& R = U.rbs[j];
                R.k = hp.rb_k[j];
                R.dil = hp.rb_d[j];
                if (R.k <= 0 || !(R.k & 1)) {
                    err = "resblock kernel sizes must be odd";
                    return false;
                }
                const std::string rb = "decoder.resblocks." + std::to_string(i * hp.rb_k.size() + j) + ".";
                R.c1.resize(R.dil.size());
                R.c2.resize(R.dil.size());
                for (size_t d = 0; d < R.dil.size(); ++d) {
                    const std::string sd = std::to_string(d);
                    if (!pack(f, rb + "convs1." + sd + ".weight", rb + "convs1." + sd + ".bias", EPI_STD, {c, c, R.k}, R.c1[d], err)) return false;
                    if (!pack(f, rb + "convs2." + sd + ".weight", rb + "convs2." + sd + ".bias", EPI_STD, {c, c, R.k}, R.c2[d], err)) return false;
                }
            }
        }
        const TensorEntry* pw = f.find("decoder.conv_post.weight");
        if (!pw) {
            err = "[ERROR] tensor not found: decoder.conv_post.weight";
            return false;
        }
        if (pw->rank != 3 || pw->ne[1] != c || pw->ne[2] != 1 || pw->ne[0] <= 0 || pw->ne[0] > 63 || !(pw->ne[0] & 1)) {
            err = "tensor 'decoder.conv_post.weight' has shape " + shape_str(*pw) + ", expected [odd k, " + std::to_string(c) + ", 1]";
            return false;
        }
        dec_post_k_ = (int)pw->ne[0];
        dec_post_cin_ = (int)pw->ne[1];
        if (!(dec_post_w_ = upload_tensor(f, "decoder.conv_post.weight", err, {dec_post_k_, dec_post_cin_, 1}))) return false;
        // one-sided receptive field of the vocoder, walked from the waveform back to the frames: conv_post, then per stage
        // the deepest resblock chain (k/2 * (d + 1) per conv pair) and the transposed conv (K taps over stride s)
        int h = dec_post_k_ / 2;
        for (int i = (int)ups_.size() - 1; i >= 0; --i) {
            int reach = 0;
            for (const ResBlockW& R : ups_[i].rbs) {
                int r = 0;
                for (int d : R.dil) r += (R.k / 2) * (d + 1);
                reach = std::max(reach, r);
            }
            h = (h + reach + ups_[i].k + ups_[i].stride - 1) / ups_[i].stride + 1;
        }
        halo_frames_ = h + dec_pre_.kt / 2 + 1;
        vocoder_group_ok_ = (hp.flow_size % 8 == 0) && (hp.up_init % 8 == 0);
        for (const UpStageW& U : ups_) vocoder_group_ok_ = vocoder_group_ok_ && (U.channels % 8 == 0);
    }
    if (!dry_run_ && hipDeviceSynchronize() != hipSuccess) {
        err = "device error while uploading weights";
        return false;
    }
    if (!dry_run_) {
        // The emulated-ggml mode (vits_model_set_ggml_tables(model, 1)) runs stage one in the exact order of include/vits_exact_math.h on the tensors AS
        // THE FILE HOLDS THEM (torch layout, no packing, no folds): keep the text encoder's and the duration predictor's, in their storage type
        // (~20 MB of host memory for the MMS-TTS architecture), and upload them when the mode is first switched on.
        for (const TensorEntry& t : f.tensors) {
            const bool s1 = t.name.rfind("text_encoder.", 0) == 0 || t.name.rfind("duration_predictor.", 0) == 0;
            if (!s1 || t.name.find(".post_") != std::string::npos || t.name.rfind("duration_predictor.flows.1.", 0) == 0) continue;
            exact_src_.push_back(t);
        }
        if (knobs.lat16_eager && ensure_lat16(err)) return false;
    }
    return true;
}

// conv_lat16_kernel's / stage1_lat.hip's A fragments (repack_conv_weights_l16) for every layer that may run on them: a second arrangement of the packed fp32
// weights, built from the device copy (download, re-order on the host, upload: ~0.2 s for the full architecture, once per handle). Called at the first call small
// enough to use the latency kernels; until then wp_l16 is null and the tile rules simply do not pick them (same bits either way).
int Engine::ensure_lat16(std::string& err) {
    if (lat16_ready_) return 0;
    std::vector<float> host;
    for (const Lat16Lazy& e : lat16_lazy_) {
        if (e.pc->wp_l16 || !e.pc->wp) continue;
        host.resize(e.n);
        HIP_OK(hipMemcpy(host.data(), e.pc->wp, e.n * sizeof(float), hipMemcpyDeviceToHost));
        float* d = upload(repack_conv_weights_l16(host, e.pc->mtiles, e.pc->nchunks, e.pc->kt));
        if (!d) {
            err = "hipMalloc failed for the latency kernels' weight copy";
            return -1;
        }
        e.pc->wp_l16 = d;
    }
    lat16_ready_ = true;
    return 0;
}

bool Engine::validate(const uint8_t* bytes, size_t size, std::string& err) {
    dry_run_ = true;
    const bool ok = load(bytes, size, err);
    owned_.clear();  // (dry-run "pointers" are not allocations)
    return ok;
}


void Engine::clear_taps() {
    for (auto& kv : taps_)
        if (kv.second.dev) hipFree(kv.second.dev);
    taps_.clear();
}

void Engine::snapshot(const char* name, TensorRef t, int channels, int stride, int batch, const std::vector<int>& lens) {
    prof.fence();
    Tap tp;
    tp.channels = channels;
    tp.stride = stride;
    tp.lens = lens;
    const size_t n = (size_t)batch * channels * stride;
    if (hipMalloc((void**)&tp.dev, n * sizeof(float)) != hipSuccess) return;
    // gather [b][c][0:stride] rows out of the (possibly wider) source tensor
    hipMemcpy2DAsync(tp.dev, (size_t)stride * 4, t.p, (size_t)t.cs * 4, (size_t)stride * 4, (size_t)channels, hipMemcpyDeviceToDevice, stream);
    for (int b = 1; b < batch; ++b)
        hipMemcpy2DAsync(tp.dev + (size_t)b * channels * stride, (size_t)stride * 4, t.p + (size_t)b * t.bs, (size_t)t.cs * 4, (size_t)stride * 4, (size_t)channels,
                         hipMemcpyDeviceToDevice, stream);
    taps_[name] = tp;
}

int64_t Engine::get_tap(const char* name, int utt, float* dst, size_t cap) {
    auto it = taps_.find(name);
    if (it == taps_.end() || utt < 0 || utt >= tap_batch_) return 0;
    const Tap& tp = it->second;
    const int len = tp.lens[utt];
    const int64_t n = (int64_t)tp.channels * len;
    if (dst && cap) {
        hipStreamSynchronize(stream);
        std::vector<float> host((size_t)tp.channels * tp.stride);
        hipMemcpy(host.data(), tp.dev + (size_t)utt * tp.channels * tp.stride, host.size() * 4, hipMemcpyDeviceToHost);
        size_t w = 0;
        for (int c = 0; c < tp.channels && w < cap; ++c)
            for (int t = 0; t < len && w < cap; ++t) dst[w++] = host[(size_t)c * tp.stride + t];
    }
    return n;
}

int Engine::set_arith(int a, std::string& err) {
    if (a == arith) return 0;
    if (a == VITS_ARITH_F32_SPLIT) {
        // two bf16 planes of A fragments for every conv the split kernels may take (conv_split.hip conv_split_candidate), built once; a conv whose weights are
        // not exactly two bf16 pieces (fp32-stored) simply keeps its fp32 kernels. Transactional like the 16-bit packing below.
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<std::pair<PackedConv*, uint16_t*>> fresh;
        std::vector<int64_t> fresh_bytes;
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < packs_.size() && e == hipSuccess; ++i) {
            const PackSrc& ps = packs_[i];
            if (ps.pc->wps || !conv_split_candidate(ps.epi, ps.k, ps.cin, ps.cout)) continue;
            const std::vector<float> w = ps.widen();
            std::vector<uint16_t> packed;
            if (!pack_conv_weights_split(w.data(), ps.cout, ps.cin, ps.k, packed)) continue;
            uint16_t* d = nullptr;
            e = hipMalloc((void**)&d, packed.size() * sizeof(uint16_t));
            if (e == hipSuccess) e = hipMemcpy(d, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            if (d) fresh.emplace_back(ps.pc, d), fresh_bytes.push_back((int64_t)packed.size() * 2);
        }
        if (e != hipSuccess) {
            for (auto& f : fresh) hipFree(f.second);
            err = std::string("vits_model_set_arith: ") + hipGetErrorString(e) + " while packing the split weight planes; the model stays in its previous arithmetic";
            return -1;
        }
        for (size_t i = 0; i < fresh.size(); ++i) {
            fresh[i].first->wps = fresh[i].second;
            fresh[i].first->bytes_s = fresh_bytes[i];
            owned_.push_back(fresh[i].second);
        }
        arith = a;
        return 0;
    }
    if (a != VITS_ARITH_F32) {
        // pack every conv's weights as 16-bit A fragments of the requested type (rounded to nearest even; a no-op on the values
        // when the file already stores that type, as the reference's exporter does for fp16: export_vits.py:87).
        // Transactional: all new buffers are built first; the model switches (pointers AND `arith`) only when every upload
        // succeeded, so a failure part-way (out of memory) leaves the previous mode fully intact.
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<uint16_t*> fresh(packs_.size(), nullptr);
        std::vector<int64_t> fresh_bytes(packs_.size(), 0);
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < packs_.size() && e == hipSuccess; ++i) {
            const PackSrc& ps = packs_[i];
            const std::vector<float> w = ps.widen();
            const std::vector<uint16_t> packed = pack_conv_weights16(w.data(), ps.cout, ps.cin, ps.k, ps.epi, ps.ct_stride, a);
            e = hipMalloc((void**)&fresh[i], packed.size() * sizeof(uint16_t));
            if (e == hipSuccess) e = hipMemcpy(fresh[i], packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            fresh_bytes[i] = (int64_t)packed.size() * 2;
        }
        if (e != hipSuccess) {
            for (uint16_t* d : fresh)
                if (d) hipFree(d);
            err = std::string("vits_model_set_arith: ") + hipGetErrorString(e) + " while packing the 16-bit weight fragments; the model stays in its previous arithmetic";
            return -1;
        }
        for (size_t i = 0; i < packs_.size(); ++i) {
            PackSrc& ps = packs_[i];
            if (ps.pc->wp16) hipFree(ps.pc->wp16);
            ps.pc->wp16 = fresh[i];
            ps.pc->bytes16 = fresh_bytes[i];
        }
    }
    arith = a;
    return 0;
}


}  // namespace vits
