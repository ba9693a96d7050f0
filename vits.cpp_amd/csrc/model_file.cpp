// model_file.cpp — reader/writer for the reference's model format, hyper-parameter parsing, synthetic models.
// See model_file.h for the format and the reference lines restated.
#include "model_file.h"

#include <cmath>
#include <cstring>
#include <sstream>

#include "../../include/vits.h"
#include "../../include/vits_synth_noise.h"

namespace vits {

// ---- fp16 / bf16 ----------------------------------------------------------------------------------
uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t man = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));  // inf / nan
    if (exp >= 31) return (uint16_t)(sign | 0x7C00u);                                         // overflow -> inf
    if (exp <= 0) {                                                                           // subnormal / zero
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;  // 14..24
        uint32_t half = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;  // may carry into exponent (correct)
    return (uint16_t)(sign | half);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) {
                man <<= 1;
                exp--;
            }
            man &= 0x3FF;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31)
        f = sign | 0x7F800000u | (man << 13);
    else
        f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}

uint16_t f32_to_bf16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    if ((x & 0x7F800000u) == 0x7F800000u) return (uint16_t)(x >> 16);
    x += 0x7FFFu + ((x >> 16) & 1);
    return (uint16_t)(x >> 16);
}
float bf16_to_f32(uint16_t h) {
    uint32_t x = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

std::vector<float> TensorEntry::to_f32() const {
    const int64_t n = count();
    std::vector<float> out((size_t)n);
    if (dtype == DT_F32) std::memcpy(out.data(), raw.data(), (size_t)n * 4);
    else {
        for (int64_t i = 0; i < n; ++i) {
            uint16_t v;
            std::memcpy(&v, raw.data() + 2 * i, 2);
            out[(size_t)i] = dtype == DT_F16 ? f16_to_f32(v) : bf16_to_f32(v);
        }
    }
    return out;
}

// ---- parse / serialize ----------------------------------------------------------------------------
namespace {
struct Rd {
    const uint8_t* p;
    size_t n, off = 0;
    bool ok = true;
    uint32_t u32() {
        if (off + 4 > n) {
            ok = false;
            return 0;
        }
        uint32_t v = (uint32_t)p[off] | ((uint32_t)p[off + 1] << 8) | ((uint32_t)p[off + 2] << 16) | ((uint32_t)p[off + 3] << 24);
        off += 4;
        return v;
    }
    std::string str() {
        uint32_t len = u32();
        if (!ok || off + len > n) {
            ok = false;
            return {};
        }
        std::string s((const char*)p + off, len);
        off += len;
        return s;
    }
};
void put_u32(std::vector<uint8_t>& o, uint32_t v) {
    for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i)));
}
void put_str(std::vector<uint8_t>& o, const std::string& s) {
    put_u32(o, (uint32_t)s.size());
    o.insert(o.end(), s.begin(), s.end());
}
}  // namespace

bool ModelFile::parse(const uint8_t* bytes, size_t size, std::string& err) {
    Rd r{bytes, size};
    vocab.clear();
    config.clear();
    tensors.clear();
    index_.clear();
    const uint32_t nv = r.u32();  // ref: vits_tokenizer.cpp:25-38
    for (uint32_t i = 0; i < nv && r.ok; ++i) {
        std::string k = r.str();
        uint32_t id = r.u32();
        vocab.emplace_back(std::move(k), id);
    }
    add_blank = r.u32();  // ref: vits_tokenizer.cpp:41-42
    normalize = r.u32();
    pad_token = r.str();  // ref: :45-52
    unk_token = r.str();
    const uint32_t nc = r.u32();  // ref: vits_model_data.cpp:36-54
    for (uint32_t i = 0; i < nc && r.ok; ++i) {
        std::string k = r.str();
        std::string v = r.str();
        config.emplace_back(std::move(k), std::move(v));
    }
    const uint32_t nt = r.u32();  // ref: vits_model_data.cpp:56-89
    for (uint32_t i = 0; i < nt && r.ok; ++i) {
        TensorEntry t;
        t.name = r.str();
        t.dtype = r.u32();
        t.rank = r.u32();
        if (!r.ok) break;
        if (t.rank > 4) {
            err = "tensor '" + t.name + "': rank > 4";
            return false;
        }
        // element count with an overflow guard: four u32 dimensions can wrap a 64-bit product, and nbytes is a u32, so a
        // count above 2^32 can never match the payload anyway
        uint64_t cnt = 1;
        bool too_big = false;
        for (uint32_t j = 0; j < t.rank; ++j) {
            t.ne[j] = r.u32();
            cnt *= (uint64_t)t.ne[j];  // cnt <= 2^32 before, factor < 2^32: no wrap
            if (cnt > ((uint64_t)1 << 32)) too_big = true, cnt = ((uint64_t)1 << 32) + 1;
        }
        const uint32_t nbytes = r.u32();
        if (r.ok && too_big) {
            err = "tensor '" + t.name + "': shape overflows the payload size";
            return false;
        }
        if (!r.ok || r.off + nbytes > r.n) {
            err = "truncated tensor '" + t.name + "'";
            return false;
        }
        const int64_t esz = t.dtype == DT_F32 ? 4 : (t.dtype == DT_F16 || t.dtype == DT_BF16) ? 2 : 0;
        if (esz == 0) {
            err = "Unsupported tensor type";  // ref: vits_model_data.cpp:85
            return false;
        }
        if ((int64_t)nbytes != esz * t.count()) {
            err = "tensor '" + t.name + "': byte length does not match shape";
            return false;
        }
        t.raw.assign(r.p + r.off, r.p + r.off + nbytes);
        r.off += nbytes;
        tensors.push_back(std::move(t));
    }
    if (!r.ok) {
        err = "truncated model file";
        return false;
    }
    return true;
}

std::vector<uint8_t> ModelFile::serialize() const {
    std::vector<uint8_t> o;
    size_t total = 64;
    for (auto& t : tensors) total += t.raw.size() + t.name.size() + 32;
    o.reserve(total + 4096);
    put_u32(o, (uint32_t)vocab.size());  // export_vits.py:11-17
    for (auto& kv : vocab) {
        put_str(o, kv.first);
        put_u32(o, kv.second);
    }
    put_u32(o, add_blank);  // :19-20
    put_u32(o, normalize);
    put_str(o, pad_token);  // :22-27
    put_str(o, unk_token);
    put_u32(o, (uint32_t)config.size());  // :30-38
    for (auto& kv : config) {
        put_str(o, kv.first);
        put_str(o, kv.second);
    }
    put_u32(o, (uint32_t)tensors.size());  // :41-70
    for (auto& t : tensors) {
        put_str(o, t.name);
        put_u32(o, t.dtype);
        put_u32(o, t.rank);
        for (uint32_t j = 0; j < t.rank; ++j) put_u32(o, (uint32_t)t.ne[j]);
        put_u32(o, (uint32_t)t.raw.size());
        o.insert(o.end(), t.raw.begin(), t.raw.end());
    }
    return o;
}

const TensorEntry* ModelFile::find(const std::string& name) const {
    if (index_.empty())
        for (size_t i = 0; i < tensors.size(); ++i) index_[tensors[i].name] = i;
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors[it->second];
}

std::string ModelFile::cfg(const std::string& key, const std::string& dflt) const {
    for (auto& kv : config)
        if (kv.first == key) return kv.second;
    return dflt;
}

// ---- hyper-parameters -----------------------------------------------------------------------------
namespace {
std::vector<int> parse_list(const std::string& s) {  // ref: vits.cpp:33-59
    std::vector<int> out;
    std::string cur;
    for (char c : s) {
        if (c == ' ' || c == '[' || c == ']') continue;
        if (c == ',') {
            if (!cur.empty()) out.push_back(std::stoi(cur));
            cur.clear();
        } else
            cur.push_back(c);
    }
    if (!cur.empty()) out.push_back(std::stoi(cur));
    return out;
}
std::vector<std::vector<int>> parse_list2(const std::string& full) {  // ref: vits.cpp:62-90
    std::vector<std::vector<int>> out;
    if (full.size() < 2) return out;
    const std::string s = full.substr(1, full.size() - 2);
    size_t i = 0;
    while (i < s.size()) {
        size_t end = i;
        int depth = 0;
        while (end < s.size()) {
            const char c = s[end];
            if (c == '[') depth++;
            else if (c == ']') depth--;
            else if (c == ',' && depth == 0) break;
            end++;
        }
        out.push_back(parse_list(s.substr(i, end - i)));
        i = end + 1;
    }
    return out;
}
}  // namespace

bool HParams::load(const ModelFile& f, std::string& err) {
    try {
        auto geti = [&](const char* k, int& v) {
            const std::string s = f.cfg(k);
            if (!s.empty()) v = std::stoi(s);
        };
        auto getf = [&](const char* k, float& v) {
            const std::string s = f.cfg(k);
            if (!s.empty()) v = std::stof(s);
        };
        geti("vocab_size", vocab_size);
        geti("hidden_size", hidden);
        geti("num_hidden_layers", layers);
        geti("num_attention_heads", heads);
        geti("window_size", window);
        geti("ffn_dim", ffn_dim);
        geti("ffn_kernel_size", ffn_k);
        geti("flow_size", flow_size);
        geti("prior_encoder_num_flows", n_flows);
        geti("prior_encoder_num_wavenet_layers", wn_layers);
        geti("wavenet_kernel_size", wn_k);
        geti("wavenet_dilation_rate", wn_rate);
        geti("upsample_initial_channel", up_init);
        if (!f.cfg("upsample_rates").empty()) up_rates = parse_list(f.cfg("upsample_rates"));
        if (!f.cfg("upsample_kernel_sizes").empty()) up_k = parse_list(f.cfg("upsample_kernel_sizes"));
        if (!f.cfg("resblock_kernel_sizes").empty()) rb_k = parse_list(f.cfg("resblock_kernel_sizes"));
        if (!f.cfg("resblock_dilation_sizes").empty()) rb_d = parse_list2(f.cfg("resblock_dilation_sizes"));
        getf("leaky_relu_slope", lrelu);
        getf("layer_norm_eps", ln_eps);
        geti("duration_predictor_kernel_size", dp_k);
        geti("depth_separable_num_layers", dds_layers);
        geti("duration_predictor_flow_bins", dp_bins);
        geti("duration_predictor_num_flows", dp_flows);
        {
            int tb = (int)dp_tail;  // the reference parses this key with stoi (vits.cpp:861)
            geti("duration_predictor_tail_bound", tb);
            dp_tail = (float)tb;
        }
        getf("noise_scale_duration", noise_scale_dur);
        getf("noise_scale", noise_scale);
        getf("speaking_rate", speaking_rate);
        geti("sampling_rate", sampling_rate);
        geti("speaker_embedding_size", speaker_embedding_size);
        if (!f.cfg("hidden_act").empty()) hidden_act = f.cfg("hidden_act");
        if (!f.cfg("use_stochastic_duration_prediction").empty()) stochastic_duration = f.cfg("use_stochastic_duration_prediction") == "True";
    } catch (const std::exception& e) {
        err = std::string("bad config value: ") + e.what();
        return false;
    }
    // what the reference refuses (vits.cpp:379-380,391,461,603-605,936-937,993), refused here too
    if (hidden_act != "relu") {
        err = "activation function not supported " + hidden_act;
        return false;
    }
    if (ffn_k <= 1) {
        err = "ffn_kernel_size == 1 not supported ";
        return false;
    }
    if (!stochastic_duration) {
        err = "Only stochastic duration prediction is supported";
        return false;
    }
    if (speaker_embedding_size != 0) {
        err = "speaker conditioning is not implemented (reference asserts the same, vits.cpp:461,603,936)";
        return false;
    }
    if (up_rates.size() != up_k.size() || rb_k.size() != rb_d.size() || heads <= 0 || hidden % heads != 0 || flow_size % 2 != 0) {
        err = "inconsistent hyper-parameters";
        return false;
    }
    for (size_t i = 0; i < up_rates.size(); ++i)
        if (up_k[i] != 2 * up_rates[i]) {
            err = "upsample kernel must be 2*stride (polyphase transposed conv assumes K/s == 2)";
            return false;
        }
    return true;
}

// ---- synthetic models -----------------------------------------------------------------------------
namespace {
struct Synth {
    ModelFile f;
    uint64_t seed;
    uint32_t ordinal = 0;
    bool bf16 = false;  // store the 16-bit tensors as bf16 (type tag 2, extension) instead of fp16
    // torch-shaped tensor filled with scale * N(0,1); stored fp16 or fp32
    void add(const std::string& name, std::vector<int64_t> shape, bool half, float scale, float offset = 0.f) {
        TensorEntry t;
        t.name = name;
        t.dtype = half ? (bf16 ? DT_BF16 : DT_F16) : DT_F32;
        t.rank = (uint32_t)shape.size();
        int64_t n = 1;
        for (size_t i = 0; i < shape.size(); ++i) {
            t.ne[i] = shape[shape.size() - 1 - i];  // export_vits.py:61-63 writes tensor.shape[::-1]
            n *= shape[i];
        }
        t.raw.resize((size_t)n * (half ? 2 : 4));
        const uint64_t stream = VITS_STREAM_WEIGHTS + ordinal++;
        for (int64_t i = 0; i < n; ++i) {
            const float v = offset + scale * vits_counter_normal(seed, stream, (uint64_t)i);
            if (half) {
                const uint16_t h = bf16 ? f32_to_bf16(v) : f32_to_f16(v);
                std::memcpy(t.raw.data() + 2 * i, &h, 2);
            } else
                std::memcpy(t.raw.data() + 4 * i, &v, 4);
        }
        f.tensors.push_back(std::move(t));
    }
    // conv weight [Cout][Cin][K] with gain/sqrt(fan_in), fp16 like export_vits.py:87; bias fp32
    void conv(const std::string& base, int cout, int cin, int k, float gain, bool bias = true, float fan_div = 1.f) {
        add(base + ".weight", {cout, cin, k}, true, gain / std::sqrt((float)cin * k / fan_div));
        if (bias) add(base + ".bias", {cout}, false, 0.02f);
    }
    void norm(const std::string& base, int c) {
        add(base + ".weight", {c}, false, 0.1f, 1.0f);
        add(base + ".bias", {c}, false, 0.05f);
    }
};

std::string list_str(const std::vector<int>& v) {  // python repr, like str(value) at export_vits.py:33
    std::ostringstream o;
    o << "[";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i];
    o << "]";
    return o.str();
}
}  // namespace

ModelFile make_synthetic_model(uint64_t seed, int arch_flags) {
    const int arch = arch_flags & 0xFF;
    HParams h;  // defaults == MMS-TTS
    if (arch == VITS_SYNTH_TINY) {
        h.hidden = 16;
        h.layers = 2;
        h.heads = 2;
        h.window = 2;
        h.ffn_dim = 32;
        h.flow_size = 16;
        h.n_flows = 2;
        h.wn_layers = 2;
        h.up_init = 32;
        h.up_rates = {4, 2};
        h.up_k = {8, 4};
        h.rb_k = {3, 5};
        h.rb_d = {{1, 3}, {1, 2}};
        h.dds_layers = 2;
    }
    Synth s;
    s.seed = seed;
    s.bf16 = (arch_flags & VITS_SYNTH_BF16) != 0;
    ModelFile& f = s.f;
    // tokenizer block: a 38-entry single-character vocabulary in the style of the MMS checkpoints
    {
        uint32_t id = 0;
        f.vocab.emplace_back("<pad>", id++);
        f.vocab.emplace_back(" ", id++);
        f.vocab.emplace_back("'", id++);
        f.vocab.emplace_back("-", id++);
        for (char c = 'a'; c <= 'z'; ++c) f.vocab.emplace_back(std::string(1, c), id++);
        for (char c = '0'; c <= '6'; ++c) f.vocab.emplace_back(std::string(1, c), id++);
        f.vocab.emplace_back("<unk>", id++);
        f.add_blank = 1;
        f.normalize = 1;
        f.pad_token = "<pad>";
        f.unk_token = "<unk>";
        h.vocab_size = (int)id;
    }
    // config block (keys and python-repr values as config.to_diff_dict() yields them)
    {
        auto put = [&](const std::string& k, const std::string& v) { f.config.emplace_back(k, v); };
        auto puti = [&](const std::string& k, int v) { put(k, std::to_string(v)); };
        puti("vocab_size", h.vocab_size);
        puti("hidden_size", h.hidden);
        puti("num_hidden_layers", h.layers);
        puti("num_attention_heads", h.heads);
        puti("window_size", h.window);
        put("use_bias", "True");
        puti("ffn_dim", h.ffn_dim);
        puti("ffn_kernel_size", h.ffn_k);
        puti("flow_size", h.flow_size);
        puti("spectrogram_bins", 513);
        put("hidden_act", "relu");
        put("layer_norm_eps", "1e-05");
        put("use_stochastic_duration_prediction", "True");
        puti("num_speakers", 1);
        puti("speaker_embedding_size", 0);
        puti("upsample_initial_channel", h.up_init);
        put("upsample_rates", list_str(h.up_rates));
        put("upsample_kernel_sizes", list_str(h.up_k));
        put("resblock_kernel_sizes", list_str(h.rb_k));
        {
            std::string v = "[";
            for (size_t i = 0; i < h.rb_d.size(); ++i) v += (i ? ", " : "") + list_str(h.rb_d[i]);
            put("resblock_dilation_sizes", v + "]");
        }
        put("leaky_relu_slope", "0.1");
        puti("depth_separable_channels", 2);
        puti("depth_separable_num_layers", h.dds_layers);
        puti("duration_predictor_flow_bins", h.dp_bins);
        put("duration_predictor_tail_bound", "5.0");
        puti("duration_predictor_kernel_size", h.dp_k);
        puti("duration_predictor_num_flows", h.dp_flows);
        puti("prior_encoder_num_flows", h.n_flows);
        puti("prior_encoder_num_wavenet_layers", h.wn_layers);
        puti("wavenet_kernel_size", h.wn_k);
        puti("wavenet_dilation_rate", h.wn_rate);
        put("speaking_rate", "1.0");
        put("noise_scale", "0.667");
        put("noise_scale_duration", "0.8");
        puti("sampling_rate", h.sampling_rate);
        put("model_type", "vits");
    }
    const int H = h.hidden, hd = H / h.heads, F = h.flow_size;
    // text encoder (HF state_dict order)
    s.add("text_encoder.embed_tokens.weight", {h.vocab_size, H}, false, 1.0f / std::sqrt((float)H));
    for (int l = 0; l < h.layers; ++l) {
        const std::string b = "text_encoder.encoder.layers." + std::to_string(l) + ".";
        s.add(b + "attention.emb_rel_k", {1, 2 * h.window + 1, hd}, false, 1.0f / std::sqrt((float)hd));
        s.add(b + "attention.emb_rel_v", {1, 2 * h.window + 1, hd}, false, 1.0f / std::sqrt((float)hd));
        for (const char* p : {"k_proj", "v_proj", "q_proj", "out_proj"}) {
            s.add(b + "attention." + p + ".weight", {H, H}, false, 1.0f / std::sqrt((float)H));
            s.add(b + "attention." + p + ".bias", {H}, false, 0.02f);
        }
        s.norm(b + "layer_norm", H);
        s.conv(b + "feed_forward.conv_1", h.ffn_dim, H, h.ffn_k, 1.4f);
        s.conv(b + "feed_forward.conv_2", H, h.ffn_dim, h.ffn_k, 1.0f);
        s.norm(b + "final_layer_norm", H);
    }
    s.conv("text_encoder.project", 2 * F, H, 1, 0.5f);
    // flow
    for (int i = 0; i < h.n_flows; ++i) {
        const std::string b = "flow.flows." + std::to_string(i) + ".";
        s.conv(b + "conv_pre", H, F / 2, 1, 1.0f);
        for (int l = 0; l < h.wn_layers; ++l) {
            s.add(b + "wavenet.in_layers." + std::to_string(l) + ".bias", {2 * H}, false, 0.02f);
            s.add(b + "wavenet.in_layers." + std::to_string(l) + ".weight", {2 * H, H, h.wn_k}, true, 1.0f / std::sqrt((float)H * h.wn_k));
        }
        for (int l = 0; l < h.wn_layers; ++l) {
            const int co = l < h.wn_layers - 1 ? 2 * H : H;
            s.add(b + "wavenet.res_skip_layers." + std::to_string(l) + ".bias", {co}, false, 0.02f);
            s.add(b + "wavenet.res_skip_layers." + std::to_string(l) + ".weight", {co, H, 1}, true, 1.5f / std::sqrt((float)H));
        }
        s.conv(b + "conv_post", F / 2, H, 1, 0.5f);
    }
    // decoder (HiFiGAN)
    s.conv("decoder.conv_pre", h.up_init, F, 7, 1.0f);
    {
        int c = h.up_init;
        for (size_t i = 0; i < h.up_rates.size(); ++i) {
            // ConvTranspose1d weight [Cin][Cout][K]; K/s = 2 taps reach each output sample
            s.add("decoder.upsampler." + std::to_string(i) + ".weight", {c, c / 2, h.up_k[i]}, true, 1.3f / std::sqrt((float)c * 2.0f));
            s.add("decoder.upsampler." + std::to_string(i) + ".bias", {c / 2}, false, 0.02f);
            c /= 2;
        }
        c = h.up_init;
        for (size_t i = 0; i < h.up_rates.size(); ++i) {
            c /= 2;
            for (size_t j = 0; j < h.rb_k.size(); ++j) {
                const std::string b = "decoder.resblocks." + std::to_string(i * h.rb_k.size() + j) + ".";
                for (size_t d = 0; d < h.rb_d[j].size(); ++d) s.conv(b + "convs1." + std::to_string(d), c, c, h.rb_k[j], 1.2f);
                for (size_t d = 0; d < h.rb_d[j].size(); ++d) s.conv(b + "convs2." + std::to_string(d), c, c, h.rb_k[j], 0.5f);
            }
        }
        s.conv("decoder.conv_post", 1, c, 7, 0.35f, false);
    }
    // stochastic duration predictor (only what inference reads; the reference never touches post_* and flows.1)
    {
        const std::string dp = "duration_predictor.";
        auto dds = [&](const std::string& b) {
            for (int i = 0; i < h.dds_layers; ++i) s.conv(b + "convs_dilated." + std::to_string(i), H, 1, h.dp_k, 1.0f);
            for (int i = 0; i < h.dds_layers; ++i) s.conv(b + "convs_pointwise." + std::to_string(i), H, H, 1, 1.0f);
            for (int i = 0; i < h.dds_layers; ++i) s.norm(b + "norms_1." + std::to_string(i), H);
            for (int i = 0; i < h.dds_layers; ++i) s.norm(b + "norms_2." + std::to_string(i), H);
        };
        s.conv(dp + "conv_pre", H, H, 1, 1.0f);
        s.conv(dp + "conv_proj", H, H, 1, 1.0f);
        dds(dp + "conv_dds.");
        s.add(dp + "flows.0.translate", {2, 1}, false, 0.15f, -1.1f);  // mean log-duration ~ +0.1 => ~2 frames per id
        // log_scale non-zero so Q5 (sign of log_scale) is visible
        s.add(dp + "flows.0.log_scale", {2, 1}, false, 0.15f);
        for (int fl = 1; fl <= h.dp_flows; ++fl) {
            const std::string b = dp + "flows." + std::to_string(fl) + ".";
            s.conv(b + "conv_pre", H, 1, 1, 1.0f);
            dds(b + "conv_dds.");
            s.conv(b + "conv_proj", 3 * h.dp_bins - 1, H, 1, 2.0f * std::sqrt((float)H / 192.0f));
        }
    }
    return std::move(s.f);
}

}  // namespace vits
