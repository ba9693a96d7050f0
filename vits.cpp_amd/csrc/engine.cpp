// engine.cpp — host orchestrator (see engine.h). Compiled with hipcc as host C++.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <sstream>

#include <dlfcn.h>

namespace vits {

// ---- roctx ranges (VITS_ROCTX=1): the phases of a call as marker ranges for `rocprofv3 --marker-trace --kernel-trace` ---------
// The marker library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): no link-time dependency, and
// nothing at all happens unless the variable is set. Host-side ranges: they bracket the ENQUEUE of a phase's kernels (the call
// is asynchronous up to the one frame-count read-back), which is what a timeline viewer lines up with the kernel trace.
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        if (!std::getenv("VITS_ROCTX")) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const RoctxApi& roctx_api() {
    static const RoctxApi api;
    return api;
}
// consecutive phases of one call: phase(n) closes the previous range and opens the next; the destructor closes the last
struct RoctxPhases {
    bool open = false;
    void phase(const char* name) {
        const RoctxApi& a = roctx_api();
        if (!a.push) return;
        if (open) a.pop();
        a.push(name);
        open = true;
    }
    ~RoctxPhases() {
        if (open) roctx_api().pop();
    }
};
// a nested range (one vocoder stage)
struct RoctxRange {
    bool open = false;
    explicit RoctxRange(const char* name) {
        const RoctxApi& a = roctx_api();
        if (a.push) {
            a.push(name);
            open = true;
        }
    }
    ~RoctxRange() {
        if (open) roctx_api().pop();
    }
};
}  // namespace

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- reference noise stream (vits.cpp:31 global engine; ggml-util.h:187-199 fresh distribution per tensor) ----
static std::default_random_engine g_ref_rng;
static std::mutex g_ref_mu;
void reference_noise_seed(uint32_t seed) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    g_ref_rng.seed(seed);
}
static void reference_noise_fill(float* dst, size_t n) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    std::normal_distribution<float> dist(0.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) dst[i] = dist(g_ref_rng);
}

// ---- tokenizer (src/vits_tokenizer.cpp:57-78,182-208; deterministic longest match instead of unordered_map order, Q11) ----
void Tokenizer::init(const ModelFile& f) {
    vocab.clear();
    for (auto& kv : f.vocab) vocab.emplace_back(kv.first, (int32_t)kv.second);
    std::stable_sort(vocab.begin(), vocab.end(), [](auto& a, auto& b) { return a.first.size() > b.first.size(); });
    add_blank = f.add_blank != 0;
    blank_id = 0;
    for (auto& kv : f.vocab)
        if (kv.first == f.pad_token) blank_id = (int32_t)kv.second;  // vocab[pad_token], vits_tokenizer.cpp:201
}

std::vector<int32_t> Tokenizer::tokenize(const std::string& text) const {
    std::string s = text;
    for (auto& c : s) c = (char)std::tolower((unsigned char)c);  // :195-197
    std::vector<int32_t> toks;
    size_t i = 0;
    while (i < s.size()) {
        bool found = false;
        for (auto& kv : vocab) {
            if (!kv.first.empty() && s.compare(i, kv.first.size(), kv.first) == 0) {
                toks.push_back(kv.second);
                i += kv.first.size();
                found = true;
                break;
            }
        }
        if (!found) i++;  // unknown bytes are skipped (:72-75)
    }
    std::vector<int32_t> fin;
    if (add_blank) {  // :200-206 ; without add_blank the reference returns an empty vector
        fin.assign(toks.size() * 2 + 1, blank_id);
        for (size_t k = 0; k < toks.size(); ++k) fin[k * 2 + 1] = toks[k];
    }
    return fin;
}

// ---- profiler -----------------------------------------------------------------------------------------
hipEvent_t Profiler::get() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void Profiler::begin(const char* name, double flop, double bytes, hipStream_t s, bool chain) {
    if (!on) return;
    auto it = ids.find(name);
    int id;
    if (it == ids.end()) {
        id = (int)names.size();
        names.push_back(name);
        ids[name] = id;
        agg.emplace_back();
    } else
        id = it->second;
    const bool share = chain && last_ok && last_s == s;
    Rec r{id, share ? last_b : get(), get(), flop, bytes, share};
    if (!share) hipEventRecord(r.a, s);
    recs.push_back(r);
    last_ok = false;
}
void Profiler::end(hipStream_t s) {
    if (!on || recs.empty()) return;
    hipEventRecord(recs.back().b, s);
    last_b = recs.back().b;
    last_s = s;
    last_ok = true;
}
void Profiler::collect() {
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = agg[r.name_id];
            a.calls++;
            a.ms += ms;
            a.flop += r.flop;
            a.bytes += r.bytes;
        }
        if (!r.a_shared) pool.push_back(r.a);
        pool.push_back(r.b);
    }
    recs.clear();
    last_ok = false;
}
void Profiler::reset() {
    collect();
    for (auto& a : agg) a = Agg();
}
std::string Profiler::report() {
    collect();
    std::ostringstream o;
    o.precision(9);
    o << "{\"kernels\":[";
    bool first = true;
    for (size_t i = 0; i < names.size(); ++i) {
        if (!agg[i].calls) continue;
        o << (first ? "" : ",") << "{\"name\":\"" << names[i] << "\",\"calls\":" << agg[i].calls << ",\"ms\":" << agg[i].ms << ",\"flop\":" << agg[i].flop
          << ",\"bytes\":" << agg[i].bytes << "}";
        first = false;
    }
    o << "]}";
    return o.str();
}
Profiler::~Profiler() {
    collect();
    for (auto e : pool) hipEventDestroy(e);
}

// ---- arena ----------------------------------------------------------------------------------------------
hipError_t Arena::reserve(size_t bytes) {
    off = 0;
    if (bytes <= cap) return hipSuccess;
    if (base) hipFree(base);
    base = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    hipError_t e = hipMalloc((void**)&base, want);
    if (e == hipSuccess) cap = want;
    return e;
}
Arena::~Arena() {
    if (base) hipFree(base);
}

// ---- load -------------------------------------------------------------------------------------------------
Engine::~Engine() {
    if (stream) hipStreamSynchronize(stream);
    for (hipStream_t s : side_)
        if (s) hipStreamSynchronize(s);
    clear_taps();
    if (!dry_run_) {
        for (void* p : owned_) hipFree(p);
        for (PackSrc& ps : packs_)
            if (ps.pc->wp16) hipFree(ps.pc->wp16);
    }
    if (pinned_) hipHostFree(pinned_);
    for (HStage& hs : hstage_) {
        if (hs.p) hipHostFree(hs.p);
        if (hs.ev) hipEventDestroy(hs.ev);
    }
    if (ev_fork_) hipEventDestroy(ev_fork_);
    for (hipEvent_t e : ev_done_)
        if (e) hipEventDestroy(e);
    for (hipStream_t s : side_)
        if (s) hipStreamDestroy(s);
    if (stream) hipStreamDestroy(stream);
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

static bool get_conv(const ModelFile& f, const std::string& wname, std::vector<float>& w, int& cout, int& cin, int& k, std::string& err) {
    const TensorEntry* t = f.find(wname);
    if (!t) {
        err = "[ERROR] tensor not found: " + wname;
        return false;
    }
    w = t->to_f32();
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
    if (!get_conv(f, wname, w, d0, d1, k, err)) return false;
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
    if (!dry_run_) packs_.push_back(PackSrc{&out, w, cout, cin, k, epi, ct_stride});
    out.wp = upload(packed);
    out.bias = bias.empty() ? nullptr : upload(bias);
    out.bytes = (int64_t)packed.size() * 4;
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
    if (const char* e = std::getenv("VITS_RB_STREAMS")) rb_streams_ = std::atoi(e) >= 2 ? 3 : 1;
    if (const char* e = std::getenv("VITS_LRELU_COPY_MINC")) lrelu_copy_minc_ = std::atoi(e);
    if (rb_streams_ > 1 && !dry_run_) {
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
    return true;
}

bool Engine::validate(const uint8_t* bytes, size_t size, std::string& err) {
    dry_run_ = true;
    const bool ok = load(bytes, size, err);
    owned_.clear();  // (dry-run "pointers" are not allocations)
    return ok;
}

// ---- forward ------------------------------------------------------------------------------------------------
// One 16-bit-operand convolution launch (conv16.hip), profiled like conv(): tile names carry a capital T
hipError_t Engine::conv16(const char* name, const PackedConv& w, const Conv16Call& c, hipStream_t stream, double bytes) {
    if (prof.on) {
        const int ncols = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
        const int tile = c.tile >= 0 ? c.tile : choose_conv16_tile(w.rows, w.epi, ncols, w.mtiles_used, c.batch);
        const bool group = c.yg || c.y16.p;
        char full[160];
        std::snprintf(full, sizeof(full), "%s|k%d|d%d|T%d|e%d%s|c%dx%d", name, w.kt, w.epi == EPI_CONVT ? -1 : (w.kt == 1 ? 1 : c.dil), tile, w.epi, group ? "g" : "", w.cin,
                      w.cout);
        const int64_t tot_in = c.sum_in >= 0 ? c.sum_in : (int64_t)c.batch * c.t_in;
        const int64_t tot_out = c.sum_out >= 0 ? c.sum_out : (int64_t)c.batch * c.t_out;
        const int64_t cols = w.epi == EPI_CONVT ? tot_in : tot_out;
        prof.begin(full, 2.0 * (double)w.rows * (double)w.cin * (double)w.kt * (double)cols, bytes, stream, /*chain=*/true);
    }
    hipError_t e = launch_conv16(w, c, arith, stream);
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
    hipError_t e = launch_to_group16(c.x, c.len_in, c.batch, w.cin, c.t_in, c.pre_act ? c.slope : 1.0f, x16, arith, stream);
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
    if (arith != VITS_ARITH_F32 && w.wp16) return conv16_transparent(name, w, c, stream);
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

#define KPROF(name, call)              \
    do {                               \
        prof.begin(name, 0, 0, stream); \
        hipError_t e__ = (call);       \
        prof.end(stream);              \
        if (e__ != hipSuccess) return e__; \
    } while (0)

// DDS block (vits.cpp:646-692): x is updated in place; y, p are scratch [B][H][ts]
hipError_t Engine::run_dds(const DdsW& d, TensorRef x, TensorRef y, TensorRef p, const int* lens, int batch, int tmax, int64_t sum_t) {
    const int H = hp.hidden;
    TensorRef none;
    int dil = 1;
    // Each layer as ONE kernel (misc_kernels.hip dds_layer_kernel; bit-identical to the three launches below). A fused block reads a
    // halo of its neighbours' columns, so a layer never writes the buffer it reads: x -> y -> p -> ... -> x.
    const int n = hp.dds_layers;
    bool fuse = n >= 2 && std::getenv("VITS_NO_DDS_FUSE") == nullptr;
    for (int i = 0, dl = 1; i < n && fuse; ++i, dl *= hp.dp_k) fuse = dds_layer_supported(d.pw[i], H, hp.dp_k, dl, arith);
    if (fuse) {
        TensorRef src = x;
        for (int i = 0; i < n; ++i) {
            TensorRef dst = i == n - 1 ? x : (src.p == y.p ? p : y);
            prof.begin("dds_layer_fused", 2.0 * H * H * (double)sum_t, 8.0 * H * (double)sum_t + (double)(arith == VITS_ARITH_F32 ? d.pw[i].bytes : d.pw[i].bytes16), stream);
            hipError_t e = launch_dds_layer(src, dst, d.dw_w[i], d.dw_b[i], d.n1_g[i], d.n1_b[i], d.pw[i], d.n2_g[i], d.n2_b[i], lens, batch, H, tmax, hp.dp_k, dil, 1e-5f,
                                            arith, stream);
            prof.end(stream);
            if (e != hipSuccess) return e;
            src = dst;
            dil *= hp.dp_k;
        }
        return hipSuccess;
    }
    for (int i = 0; i < hp.dds_layers; ++i) {
        KPROF("dds_depthwise_ln_gelu", launch_dds_depthwise(x, none, d.dw_w[i], d.dw_b[i], d.n1_g[i], d.n1_b[i], y, lens, batch, H, tmax, hp.dp_k, dil, 1e-5f, stream, arith));
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
        KPROF("ln_gelu_residual", launch_add_layer_norm(p, none, d.n2_g[i], d.n2_b[i], none, lens, batch, H, tmax, 1e-5f, 1, x, stream));
        dil *= hp.dp_k;
    }
    return hipSuccess;
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
    if (a != VITS_ARITH_F32) {
        // pack every conv's weights as 16-bit A fragments of the requested type (rounded to nearest even; a no-op on the values
        // when the file already stores that type, as the reference's exporter does for fp16: export_vits.py:87)
        HIP_OK(hipStreamSynchronize(stream));
        for (PackSrc& ps : packs_) {
            const std::vector<uint16_t> packed = pack_conv_weights16(ps.w.data(), ps.cout, ps.cin, ps.k, ps.epi, ps.ct_stride, a);
            uint16_t* d = nullptr;
            HIP_OK(hipMalloc((void**)&d, packed.size() * sizeof(uint16_t)));
            HIP_OK(hipMemcpy(d, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            if (ps.pc->wp16) hipFree(ps.pc->wp16);
            ps.pc->wp16 = d;
            ps.pc->bytes16 = (int64_t)packed.size() * 2;
        }
    }
    arith = a;
    return 0;
}

int Engine::sync(std::string& err) {
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
}

int Engine::process_batch(const int32_t* ids, const int32_t* id_lens, int B, int id_stride, const vits_process_opts& o, vits_batch_result* out,
                          std::string& err) {
    if (B <= 0 || id_stride <= 0) {
        err = "empty batch";
        return -1;
    }
    if (o.on_chunk && o.skip_host_copy) {  // (the only misuse of the sink; checked before any work is queued)
        err = "on_chunk needs a host copy (skip_host_copy = 0)";
        return -1;
    }
    const int md = o.mode == VITS_MODE_DEFAULT ? mode : o.mode;
    const bool refmode = md == VITS_MODE_REFERENCE;
    const int H = hp.hidden, F = hp.flow_size, heads = hp.heads, hd = H / heads;
    std::vector<int> tlen(B);
    int Tmax = 0;
    for (int b = 0; b < B; ++b) {
        tlen[b] = id_lens ? id_lens[b] : id_stride;
        if (tlen[b] <= 0 || tlen[b] > id_stride) {
            err = "bad id length";
            return -1;
        }
        Tmax = std::max(Tmax, tlen[b]);
        for (int t = 0; t < tlen[b]; ++t) {
            const int id = ids[(size_t)b * id_stride + t];
            if (id < 0 || id >= hp.vocab_size) {
                err = "token id out of range";
                return -1;
            }
        }
    }
    if (Tmax > 2048) {
        err = "more than 2048 ids per utterance is not supported";
        return -1;
    }
    const bool want_async = o.async && o.skip_host_copy && o.fixed_duration > 0 && !o.collect_taps;
    const int ts = round_up(Tmax, 32);
    const int n_up = (int)ups_.size();
    clear_taps();
    tap_batch_ = B;

    // ---- stage one buffers ------------------------------------------------------------------------------
    struct S1 {
        int *ids, *lens, *cum, *frames, *stage_lens, *stage_mul, *stage_add, *seed_off;
        float *x, *qkv, *att, *tmp, *ffn, *stats, *dpx, *dpy, *dpp, *cond, *z, *u, *dur;
        uint16_t* x16;
    } s1;
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
        // 16-bit arithmetic modes: scratch for the rounded copy of a conv input (largest c_in of stage one)
        x16_elems1 = arith != VITS_ARITH_F32 ? (size_t)B * round_up(std::max({hp.ffn_dim, 2 * F, H}), 8) * round_up(ts, 8) : 0;
        s1.x16 = x16_elems1 ? a.alloc<uint16_t>(x16_elems1) : nullptr;
    };
    {
        Arena measure;
        measure.cap = (size_t)1 << 60;
        layout1(measure);
        const size_t need = measure.off + 4096;
        measure.cap = 0;
        if (need > a1_.cap) HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(a1_.reserve(need));
        layout1(a1_);
        for (int i = 0; i < 3; ++i) {
            x16_[i] = Ref16();
            x16_cap_[i] = 0;
        }
        x16_[0].p = s1.x16;
        x16_cap_[0] = x16_elems1;
    }
    auto TR = [](float* p, int channels, int stride) {
        TensorRef t;
        t.p = p;
        t.cs = stride;
        t.bs = (int64_t)channels * stride;
        return t;
    };
    TensorRef none;
    // vocoder stage lengths as affine functions of the frame count L: len_i = L*mul_i + add_i (Q1: the reference
    // never crops the transposed conv, so every stage gains K - s samples; vits.cpp:187)
    std::vector<int> smul(n_up + 1), sadd(n_up + 1);
    smul[0] = 1;
    sadd[0] = 0;
    for (int i = 0; i < n_up; ++i) {
        const int s = ups_[i].stride, K = ups_[i].k;
        const int crop = refmode ? 0 : (K - s) / 2;
        smul[i + 1] = smul[i] * s;
        sadd[i + 1] = sadd[i] * s + (K - s - 2 * crop);
    }
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

    const int* dl = s1.lens;
    int64_t sum_t = 0;  // (profiler accounting: real work = sum of the utterance lengths)
    for (int b = 0; b < B; ++b) sum_t += tlen[b];
    TensorRef x = TR(s1.x, H, ts), qkv = TR(s1.qkv, 3 * H, ts), att = TR(s1.att, H, ts), tmp = TR(s1.tmp, H, ts), ffn = TR(s1.ffn, hp.ffn_dim, ts);
    auto sub = [](TensorRef t, int c0) {
        t.p += (int64_t)c0 * t.cs;
        return t;
    };
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

    // ---- text encoder (vits.cpp:244-440) ---------------------------------------------------------------------
    RoctxPhases rx;
    rx.phase("vits.text_encoder");
    prof.begin("embed", 0, 0, stream);
    HIP_OK(launch_embed(s1.ids, id_stride, dl, emb_, H, (float)std::sqrt((double)H), x, B, Tmax, stream));
    prof.end(stream);
    const float q_scale = (float)std::pow((double)hd, -0.5);
    for (int l = 0; l < hp.layers; ++l) {
        const EncoderLayerW& L = enc_[l];
        HIP_OK(conv("enc_qkv_gemm", L.qkv, mk(x, qkv, Tmax)));
        prof.begin("rel_attention", 0, 0, stream);
        HIP_OK(launch_rel_attention(sub(qkv, 0), sub(qkv, H), sub(qkv, 2 * H), L.rel_k, L.rel_v, att, dl, B, heads, hd, Tmax, hp.window, q_scale, stream));
        prof.end(stream);
        {
            ConvCall c = mk(att, tmp, Tmax);
            c.res = x;  // residual + attention output (vits.cpp:367)
            HIP_OK(conv("enc_out_gemm", L.out, c));
        }
        prof.begin("layer_norm", 0, 0, stream);
        HIP_OK(launch_add_layer_norm(tmp, none, L.ln1_g, L.ln1_b, x, dl, B, H, Tmax, hp.ln_eps, 0, none, stream));
        prof.end(stream);
        {
            ConvCall c = mk(x, ffn, Tmax);
            c.pad_l = (hp.ffn_k - 1) / 2;  // vits.cpp:388
            c.post_act = 1;                // relu :397
            HIP_OK(conv("enc_ffn_conv", L.ffn1, c));
        }
        {
            ConvCall c = mk(ffn, tmp, Tmax);
            c.pad_l = (hp.ffn_k - 1) / 2;
            c.res = x;  // :416
            HIP_OK(conv("enc_ffn_conv", L.ffn2, c));
        }
        prof.begin("layer_norm", 0, 0, stream);
        HIP_OK(launch_add_layer_norm(tmp, none, L.ln2_g, L.ln2_b, x, dl, B, H, Tmax, hp.ln_eps, 0, none, stream));
        prof.end(stream);
    }
    TensorRef stats = TR(s1.stats, 2 * F, ts);
    HIP_OK(conv("enc_project", enc_proj_, mk(x, stats, Tmax)));  // :429 ; split :436 = channel ranges [0,F) and [F,2F)
    if (o.collect_taps) {
        snapshot("enc_out", x, H, Tmax, B, tlen);
        snapshot("prior_mean", sub(stats, 0), F, Tmax, B, tlen);
        snapshot("prior_logvar", sub(stats, F), F, Tmax, B, tlen);
    }

    // ---- stochastic duration predictor, reverse (vits.cpp:927-972) ----------------------------------------
    rx.phase("vits.duration_predictor");
    TensorRef dpx = TR(s1.dpx, H, ts), dpy = TR(s1.dpy, H, ts), dpp = TR(s1.dpp, H, ts), cond = TR(s1.cond, H, ts), z = TR(s1.z, 2, ts), u = TR(s1.u, 32, ts);
    HIP_OK(conv("conv1x1_dp", dp_pre_, mk(x, dpx, Tmax)));
    HIP_OK(run_dds(dp_dds_, dpx, dpy, dpp, dl, B, Tmax, sum_t));
    HIP_OK(conv("conv1x1_dp", dp_proj_, mk(dpx, cond, Tmax)));
    std::vector<float> host_noise;
    if (o.noise_kind == VITS_NOISE_COUNTER) {
        prof.begin("noise_dur", 0, 0, stream);
        HIP_OK(launch_noise_dur(z, dl, B, Tmax, o.noise_seed, s1.seed_off, hp.noise_scale_dur, stream));
        prof.end(stream);
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
            // conv_pre (1 -> H, vits.cpp:864) fused with "inputs + global_conditioning" of the DDS block (:651-653)
            prof.begin("dp_flow_pre", 0, 0, stream);
            HIP_OK(launch_pointwise_from1(z, c_first, W.pre_w, W.pre_b, cond, dpy, dl, B, H, Tmax, stream, arith));
            prof.end(stream);
            HIP_OK(run_dds(W.dds, dpy, dpx, dpp, dl, B, Tmax, sum_t));
            HIP_OK(conv("conv1x1_dp", W.proj, mk(dpy, u, Tmax)));
            prof.begin("dp_spline", 0, 0, stream);
            HIP_OK(launch_spline(u, z, 1 - c_first, dl, B, Tmax, hp.dp_bins, hp.dp_tail, inv_sqrt, md, stream));
            prof.end(stream);
        }
    }
    if (o.collect_taps) snapshot("log_duration", sub(z, c_first), 1, Tmax, B, tlen);
    prof.begin("durations", 0, 0, stream);
    HIP_OK(launch_durations(z, c_first, dl, B, id_stride, (float)(1.0 / hp.speaking_rate), o.fixed_duration, s1.dur, s1.cum, s1.frames, s1.stage_lens, n_up + 1,
                            s1.stage_mul, s1.stage_add, stream));
    prof.end(stream);

    // ---- the one data-dependent shape (vits.cpp:1133): frames per utterance ---------------------------------
    rx.phase("vits.frame_count_sync");
    std::vector<int> frames(B);
    if (o.fixed_duration > 0) {
        for (int b = 0; b < B; ++b) frames[b] = std::max(1, o.fixed_duration * tlen[b]);
    } else {
        HIP_OK(hipMemcpyAsync(frames.data(), s1.frames, sizeof(int) * B, hipMemcpyDeviceToHost, stream));
        prof.fence();
        HIP_OK(hipStreamSynchronize(stream));
        prof.fence();
    }
    int Lmax = 0;
    for (int b = 0; b < B; ++b) Lmax = std::max(Lmax, frames[b]);
    std::vector<std::vector<int>> slen(n_up + 1, std::vector<int>(B));
    std::vector<int> smax(n_up + 1, 0);
    for (int i = 0; i <= n_up; ++i)
        for (int b = 0; b < B; ++b) {
            slen[i][b] = frames[b] * smul[i] + sadd[i];
            smax[i] = std::max(smax[i], slen[i][b]);
        }
    if (o.frames_only) {
        // dispatcher query: predicted frames / samples per utterance, no audio (buffer sizing, shard balancing by frames)
        if (out) {
            out->batch = (size_t)B;
            out->stride = (size_t)smax[n_up];
            out->lengths = new int64_t[B];
            out->frames = new int64_t[B];
            out->data = nullptr;
            for (int b = 0; b < B; ++b) {
                out->lengths[b] = slen[n_up][b];
                out->frames[b] = frames[b];
            }
        }
        HIP_OK(hipStreamSynchronize(stream));
        prof.fence();
        return 0;
    }
    if (o.collect_taps) {
        TensorRef d;
        d.p = s1.dur;
        d.cs = id_stride;
        d.bs = id_stride;
        snapshot("durations", d, 1, Tmax, B, tlen);
    }

    // ---- vocoder windows (long-form / streaming, vits.h vocoder_chunk_frames) ------------------------------------
    // Window w owns frames [f0, f1) and computes frames [lo, hi) = the owned range widened by the vocoder's receptive
    // field (halo_frames_): every sample it emits sees exactly the inputs it sees in a whole-utterance run, in the same
    // order of accumulation, so the PCM is bit-identical while the activations are bounded by the window.
    struct Win {
        int f0, f1, lo, hi;
    };
    std::vector<Win> wins;
    {
        const int W = (o.vocoder_chunk_frames > 0 && !o.collect_taps && o.vocoder_chunk_frames < Lmax) ? o.vocoder_chunk_frames : 0;
        if (!W) wins.push_back({0, Lmax, 0, Lmax});
        else
            for (int f0 = 0; f0 < Lmax; f0 += W) {
                const int f1 = std::min(Lmax, f0 + W);
                wins.push_back({f0, f1, std::max(0, f0 - halo_frames_), std::min(Lmax, f1 + halo_frames_)});
            }
    }
    const bool windowed = wins.size() > 1;
    // (a run that is not windowed — no chunking asked for, a window at least as long as the longest utterance, which the
    // caller cannot know in advance, or collect_taps — streams as ONE window: the sink gets each utterance in a single call)
    int Lw_max = 0;
    for (const Win& w : wins) Lw_max = std::max(Lw_max, w.hi - w.lo);
    const int M = smul[n_up];  // samples per frame

    // ---- stage two buffers -------------------------------------------------------------------------------------
    const int ls = round_up(Lmax, 32), lws = round_up(Lw_max, 32);
    size_t big = 0;  // floats of the largest vocoder activation (of one window)
    std::vector<int> sts(n_up + 1);
    for (int i = 0; i <= n_up; ++i) sts[i] = round_up(Lw_max * smul[i] + sadd[i], 32);
    for (int i = 0; i < n_up; ++i) big = std::max(big, (size_t)B * ups_[i].channels * sts[i + 1]);
    struct S2 {
        float *zp, *noise, *hout, *gate, *h0, *bu, *bul, *by[3], *bt[3], *byl[3], *bs, *bs16, *pre, *wave;
        uint16_t* x16[3];
        int* win_lens;  // [window][n_up + 2][B]: stage lengths of each utterance inside the window, then its emit end
    } s2;
    const bool need_noise_buf = o.noise_kind != VITS_NOISE_COUNTER;
    // 16-bit arithmetic modes: the vocoder runs in the group layout of conv16.hip when its channel counts allow it (taps need the
    // fp32 layout of the transparent path: collect_taps keeps to that one); scratch for the transparent path's rounded inputs
    const bool fast16 = arith != VITS_ARITH_F32 && vocoder_group_ok_ && std::getenv("VITS_NO_GROUP16") == nullptr;
    const bool fuse16 = fast16 && std::getenv("VITS_NO_FUSE16") == nullptr;  // resblock conv pairs of the narrow stages as one kernel
    const size_t x16_elems2 = arith != VITS_ARITH_F32 ? std::max({big, (size_t)B * H * round_up(ls, 8), (size_t)B * hp.up_init * round_up(lws, 8), (size_t)B * round_up(F, 8) * round_up(ls, 8)}) + 64 : 0;
    const int S_stride = round_up(smax[n_up], 32);
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
            const bool own = j == 0 || (rb_streams_ > 1 && (size_t)j < hp.rb_k.size());
            s2.by[j] = own ? a.alloc<float>(big) : s2.by[0];
            s2.bt[j] = own ? a.alloc<float>(big) : s2.bt[0];
            s2.byl[j] = own ? a.alloc<float>(big) : s2.byl[0];
        }
        s2.bs = a.alloc<float>(big);
        s2.bs16 = fast16 ? a.alloc<float>(big / 2 + 64) : nullptr;
        for (int j = 0; j < 3; ++j) s2.x16[j] = (x16_elems2 && (j == 0 || (rb_streams_ > 1 && !fast16))) ? a.alloc<uint16_t>(x16_elems2) : nullptr;
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
    const int* d_len_full[8];
    for (int i = 0; i <= n_up && i < 8; ++i) d_len_full[i] = s1.stage_lens + (size_t)i * B;
    const int* const* d_len = d_len_full;  // (the vocoder loop below shadows this with window-local lengths)

    // ---- prior sampling through the alignment (vits.cpp:1028-1064) -------------------------------------------
    rx.phase("vits.prior_sampling");
    TensorRef zp = TR(s2.zp, F, ls), noise = TR(s2.noise, F, ls);
    if (need_noise_buf) {
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

    // ---- residual coupling flow, reverse (vits.cpp:519-538,500-517,452-498) ------------------------------------
    rx.phase("vits.flow");
    {
        const int* ll = d_len[0];
        int64_t sum_frames = 0;
        for (int b = 0; b < B; ++b) sum_frames += frames[b];
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
        for (int i = hp.n_flows - 1; i > -1; --i) {
            const FlowLayerW& Lw = flow_[i];
            const bool flipped = ((hp.n_flows - i) % 2) == 1;
            TensorRef x0 = sub(zp, flipped ? F / 2 : 0), x1 = sub(zp, flipped ? 0 : F / 2);
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
            bool fuse_wn = std::getenv("VITS_NO_WN_FUSE") == nullptr && (ls & 3) == 0 && (int64_t)((Lmax + 31) / 32) * B >= 384 &&
                           (reinterpret_cast<uintptr_t>(hout.p) & 15) == 0 && (reinterpret_cast<uintptr_t>(gate.p) & 15) == 0;
            for (int l = 0; l < hp.wn_layers && fuse_wn; ++l) {
                int dl = 1;
                for (int q = 0; q < l; ++q) dl *= hp.wn_rate;
                fuse_wn = (arith == VITS_ARITH_F32 ? wavenet32_supported(H, hp.wn_k, dl, Lw.in_layers[l], Lw.res_skip[l])
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
                        std::snprintf(full, sizeof(full), "flow_wavenet_layer|k%d|d1|%c%d|e1|c%dx%d", hp.wn_k, arith == VITS_ARITH_F32 ? 'w' : 'W', H, H, Lw.res_skip[l].cout);
                        prof.begin(full, 2.0 * ((double)2 * H * H * hp.wn_k + (double)Lw.res_skip[l].cout * H) * (double)sum_frames,
                                   4.0 * (double)sum_frames * (H + 2.0 * Lw.res_skip[l].cout) + (double)Lw.in_layers[l].bytes + (double)Lw.res_skip[l].bytes, stream, true);
                    }
                    if (arith == VITS_ARITH_F32) HIP_OK(launch_wavenet32(Lw.in_layers[l], Lw.res_skip[l], w, stream));
                    else HIP_OK(launch_wavenet16(Lw.in_layers[l], Lw.res_skip[l], w, arith, stream));
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
        if (o.collect_taps) snapshot("z_flow", zp, F, Lmax, B, frames);
    }

    // ---- HiFiGAN (vits.cpp:583-644) -----------------------------------------------------------------------------
    rx.phase("vits.hifigan");
    float* wave_dst = s2.wave;
    int64_t wave_stride = S_stride;
    if (o.out_device) {
        if (o.out_device_stride < smax[n_up]) {
            err = "out_device_stride is smaller than the longest utterance";
            return -1;
        }
        wave_dst = (float*)o.out_device;
        wave_stride = o.out_device_stride;
    }
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
                const Win& wn = wins[w];
                const int lf = std::min(frames[b], wn.hi) - wn.lo;
                int* row = wl.data() + w * (size_t)(n_up + 2) * B;
                if (lf <= 0) continue;
                for (int i = 0; i <= n_up; ++i) row[(size_t)i * B + b] = lf * smul[i] + sadd[i];
                if (frames[b] > wn.f0)  // owns frames here; the window holding the utterance's end also emits its tail (Q1)
                    row[(size_t)(n_up + 1) * B + b] = frames[b] <= wn.f1 ? (frames[b] - wn.lo) * M + sadd[n_up] : (wn.f1 - wn.lo) * M;
            }
        HIP_OK(hipMemcpyAsync(s2.win_lens, wl.data(), sizeof(int) * wl.size(), hipMemcpyHostToDevice, stream));
        prof.fence();
        HIP_OK(hipStreamSynchronize(stream));  // wl goes out of scope
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
        const Win& wn = wins[w];
        if (hipEventSynchronize(chunk_ev[w]) != hipSuccess) return -1;
        for (int b = 0; b < B; ++b) {
            if (frames[b] <= wn.f0) continue;
            const size_t off = (size_t)wn.f0 * M;
            const size_t end = frames[b] <= wn.f1 ? (size_t)slen[n_up][b] : (size_t)wn.f1 * M;
            if (o.on_chunk(o.on_chunk_user, b, off, host_pcm + (size_t)b * out_stride + off, end - off)) return 1;
        }
        return 0;
    };
    for (size_t wi = 0; wi < wins.size(); ++wi) {
        const Win& wn = wins[wi];
        const int Lw = wn.hi - wn.lo;
        const int* d_len[8];
        std::vector<int> smax(n_up + 1);  // (shadows the whole-utterance maxima: everything below is window-local)
        for (int i = 0; i <= n_up && i < 8; ++i) {
            d_len[i] = windowed ? s2.win_lens + (wi * (size_t)(n_up + 2) + i) * B : d_len_full[i];
            smax[i] = Lw * smul[i] + sadd[i];
        }
        std::vector<int64_t> ssum(n_up + 1, 0);  // (profiler accounting) sum over utterances of the stage lengths inside this window
        for (int b = 0; b < B; ++b) {
            const int lf = std::min(frames[b], wn.hi) - wn.lo;
            if (lf <= 0) continue;
            for (int i = 0; i <= n_up; ++i) ssum[i] += (int64_t)lf * smul[i] + sadd[i];
        }
        const int* emit_hi = windowed ? s2.win_lens + (wi * (size_t)(n_up + 2) + n_up + 1) * B : nullptr;
        const int emit_lo = (wn.f0 - wn.lo) * M;
        TensorRef zwin = zp;
        zwin.p += wn.lo;
        TensorRef pre;
        pre.p = s2.pre;
        pre.bs = S_stride;
        pre.cs = S_stride;
        TensorRef wv;
        wv.p = wave_dst + (int64_t)wn.lo * M;  // window-local sample 0 is global sample lo * M
        wv.bs = wave_stride;
        wv.cs = (int)wave_stride;
        const float final_slope = refmode ? hp.lrelu : 0.01f;  // Q2 (vits.cpp:638)
        if (fast16) {
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
            HIP_OK(launch_to_group16(zwin, d_len[0], B, F, Lw, 1.0f, z16, arith, stream));
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
                const bool par = rb_streams_ > 1 && nk >= 2 && nk <= 3 && !prof.on;
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
                    bool fuse_rb = fuse16;
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
                            HIP_OK(launch_rbpair16(R.c1[d], R.c2[d], f, arith, sj));
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
            HIP_OK(launch_conv_post16(cur16, dec_post_w_, dec_post_cin_, dec_post_k_, pre, wv, d_len[n_up], B, smax[n_up], arith, stream, emit_lo, emit_hi));
            prof.end(stream);
        } else {
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
                    if (C >= lrelu_copy_minc_) {  // activated copy for the first conv of each resblock (see below)
                        c.y2 = s2.bul;
                        c.post_slope = hp.lrelu;
                    }
                    HIP_OK(conv("hifigan_upsample_convT", U.up, c));
                }
                // resblock j runs on its own stream (engine.h); only the LAST convolution of each resblock touches the shared
                // sum, and those are chained j-1 -> j by events so the additions keep the reference's order (vits.cpp:622-635)
                // (per-kernel event timing needs kernels that do not overlap: the profiler serialises the stage)
                const bool par = rb_streams_ > 1 && nk >= 2 && nk <= 3 && !prof.on;
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
                    const bool lcopy = C >= lrelu_copy_minc_;
                    TensorRef byl = TR(s2.byl[par ? j : 0], C, sts[st_out]), bul = TR(s2.bul, C, sts[st_out]);
                    // narrow stages: each pair as ONE kernel, t stays in LDS (rbpair32.hip; bit-identical to the two launches below).
                    // A fused block reads a halo of its neighbours' input columns while other blocks store their output, so the
                    // resblock's stream ping-pongs between `by` and the buffer the two-launch path uses for t. All pairs or none.
                    bool fuse_rb = std::getenv("VITS_NO_FUSE32") == nullptr;  // (the fused kernel reads the RAW stream: the activated copies of wide stages are for the other resblocks)
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
            HIP_OK(launch_conv_post(cur, dec_post_w_, dec_post_cin_, dec_post_k_, final_slope, pre, wv, d_len[n_up], B, smax[n_up], stream, emit_lo, emit_hi, arith));
            prof.end(stream);
        }
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
            snapshot("pre_tanh", pre, 1, smax[n_up], B, slen[n_up]);
            snapshot("waveform", wv, 1, smax[n_up], B, slen[n_up]);
        }
    }
    if (o.on_chunk)
        if (int rc = deliver(wins.size() - 1)) {
            hipStreamSynchronize(stream);
            err = rc > 0 ? "aborted by the on_chunk callback" : "hipEventSynchronize failed";
            return -1;
        }

    // ---- results ------------------------------------------------------------------------------------------------
    rx.phase("vits.results");
    if (out) {
        out->batch = (size_t)B;
        out->stride = (size_t)smax[n_up];
        out->lengths = new int64_t[B];
        out->frames = new int64_t[B];
        for (int b = 0; b < B; ++b) {
            out->lengths[b] = slen[n_up][b];
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
