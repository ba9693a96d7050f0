/*
 * vits_oracle.cpp — CPU ORACLE: plain-C++ restatement of the reference's VITS inference path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see vits_oracle.h for the parity-pin statement). Every function cites the
 * reference lines it restates; `ref:` = /root/reference/, `HF:` = transformers/models/vits/modeling_vits.py
 * (the model the reference ports, ref:src/vits.cpp:113).
 *
 * Layout: every activation is a dense [channels][time] fp32 array, time fastest — the reference's ggml
 * ne order [time, channels, 1]. Arithmetic is straight fp32 loops (the compiler may contract a*b+c to fma).
 */
#include "vits_oracle.h"
#include "vits_oracle_exact.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../include/vits_synth_noise.h"

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------
// tiny thread pool (ref: ggml worker threads, src/vits.cpp:1084-1091; count rule common.h:19-21)
// ------------------------------------------------------------------------------------------------
int default_threads() { return std::max((int)std::thread::hardware_concurrency(), 6); }

// persistent workers (like ggml's pool): spawning threads per op would dominate the small ops on many-core hosts
class Pool {
  public:
    static Pool& get() {
        static Pool p;
        return p;
    }
    void run(int threads, int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
        std::unique_lock<std::mutex> run_lock(run_mu_);  // one parallel region at a time
        const int nt = (int)std::min<int64_t>(threads, n);
        resize(std::max(threads, 1) - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            chunk_ = std::max<int64_t>(1, n / (nt * 4));
            next_.store(0);
            active_ = nt - 1;
            pending_ = nt - 1;
            ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    std::mutex run_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int64_t, int64_t)>* fn_ = nullptr;
    int64_t n_ = 0, chunk_ = 1;
    std::atomic<int64_t> next_{0};
    int active_ = 0, pending_ = 0;
    int limit_ = 1 << 30;  // workers with id >= limit_ retire (the pool follows the requested thread count DOWN as well:
                           // every region wakes all parked workers, so 255 idle ones left over from a 256-thread run make a
                           // 32-thread run 1.8x slower — the sweep-vs-sustained gap of round 1's cpu_baseline)
    uint64_t epoch_ = 0;
    bool stop_ = false;
    void work() {
        for (;;) {
            const int64_t b = next_.fetch_add(chunk_);
            if (b >= n_) break;
            (*fn_)(b, std::min(n_, b + chunk_));
        }
    }
    void resize(int count) {
        if ((int)workers_.size() > count) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                limit_ = count;
            }
            cv_.notify_all();
            while ((int)workers_.size() > count) {
                workers_.back().join();
                workers_.pop_back();
            }
            std::lock_guard<std::mutex> lk(mu_);
            limit_ = 1 << 30;
        }
        while ((int)workers_.size() < count) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return stop_ || id >= limit_ || (epoch_ != seen && id < active_); });
                    if (stop_ || id >= limit_) return;
                    seen = epoch_;
                    lk.unlock();
                    work();
                    lk.lock();
                    if (--pending_ == 0) done_cv_.notify_all();
                }
            });
        }
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
};

void parallel_for(int threads, int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
    if (threads <= 1 || n <= 1) {
        fn(0, n);
        return;
    }
    Pool::get().run(threads, n, fn);
}

// ------------------------------------------------------------------------------------------------
// model file (ref: src/vits_model_data.cpp:29-97, src/vits_tokenizer.cpp:22-55; writer scripts/export_vits.py:5-70)
// ------------------------------------------------------------------------------------------------
struct Tensor {
    int dtype = 0;  // 0 f32, 1 f16 (as stored)
    int rank = 0;
    int64_t ne[4] = {1, 1, 1, 1};  // file order == ggml ne == reversed torch shape
    std::vector<float> d;          // widened to fp32, torch row-major
    int64_t n() const { return (int64_t)d.size(); }
};

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else {  // subnormal
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) {
                man <<= 1;
                exp--;
            }
            man &= 0x3FF;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7F800000u | (man << 13);
    } else {
        f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}

struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    bool ok = true;
    uint32_t u32() {
        if (off + 4 > n) {
            ok = false;
            return 0;
        }
        uint32_t v;
        std::memcpy(&v, p + off, 4);  // little-endian host (ref: common.h:13-17)
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

}  // namespace

struct vo_model {
    std::map<std::string, int32_t> vocab;
    uint32_t add_blank = 1, normalize = 1;
    std::string pad_token, unk_token;
    std::map<std::string, std::string> config;
    std::map<std::string, Tensor> tensors;

    // hyper-parameters (ref keys read at vits.cpp:246-254,453-457,501,523,585-595,648-649,858-861,930,977-979)
    int hidden = 192, layers = 6, heads = 2, window = 4, ffn_k = 3, flow_size = 192;
    int n_flows = 4, wn_layers = 4, wn_k = 5, wn_rate = 1;
    int up_init = 512;
    std::vector<int> up_rates{8, 8, 2, 2}, up_k{16, 16, 4, 4}, rb_k{3, 7, 11};
    std::vector<std::vector<int>> rb_d{{1, 3, 5}, {1, 3, 5}, {1, 3, 5}};
    float lrelu = 0.1f, ln_eps = 1e-5f;
    int dp_k = 3, dds_layers = 3, dp_bins = 10, dp_flows = 4;
    float dp_tail = 5.f, noise_scale_dur = 0.8f, noise_scale = 0.667f, speaking_rate = 1.0f;
    int sampling_rate = 16000;

    const Tensor& T(const std::string& name) const {
        auto it = tensors.find(name);
        if (it == tensors.end()) throw std::runtime_error("tensor not found: " + name);  // ref: vits_model_data.cpp:144
        return it->second;
    }
    bool has(const std::string& name) const { return tensors.count(name) != 0; }
};

namespace {

std::vector<int> parse_int_list(const std::string& s) {  // ref: vits.cpp:33-59
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
std::vector<std::vector<int>> parse_int_list2(const std::string& full) {  // ref: vits.cpp:62-90
    std::vector<std::vector<int>> out;
    std::string s = full.substr(1, full.size() - 2);
    size_t i = 0;
    while (i < s.size()) {
        size_t end = i;
        int depth = 0;
        while (end < s.size()) {
            char c = s[end];
            if (c == '[') depth++;
            else if (c == ']') depth--;
            else if (c == ',' && depth == 0) break;
            end++;
        }
        out.push_back(parse_int_list(s.substr(i, end - i)));
        i = end + 1;
    }
    return out;
}

void load_hparams(vo_model& m) {
    auto& c = m.config;
    auto geti = [&](const char* k, int& v) {
        auto it = c.find(k);
        if (it != c.end() && !it->second.empty()) v = std::stoi(it->second);
    };
    auto getf = [&](const char* k, float& v) {
        auto it = c.find(k);
        if (it != c.end() && !it->second.empty()) v = std::stof(it->second);
    };
    geti("hidden_size", m.hidden);
    geti("num_hidden_layers", m.layers);
    geti("num_attention_heads", m.heads);
    geti("window_size", m.window);
    geti("ffn_kernel_size", m.ffn_k);
    geti("flow_size", m.flow_size);
    geti("prior_encoder_num_flows", m.n_flows);
    geti("prior_encoder_num_wavenet_layers", m.wn_layers);
    geti("wavenet_kernel_size", m.wn_k);
    geti("wavenet_dilation_rate", m.wn_rate);
    geti("upsample_initial_channel", m.up_init);
    if (c.count("upsample_rates")) m.up_rates = parse_int_list(c["upsample_rates"]);
    if (c.count("upsample_kernel_sizes")) m.up_k = parse_int_list(c["upsample_kernel_sizes"]);
    if (c.count("resblock_kernel_sizes")) m.rb_k = parse_int_list(c["resblock_kernel_sizes"]);
    if (c.count("resblock_dilation_sizes")) m.rb_d = parse_int_list2(c["resblock_dilation_sizes"]);
    getf("leaky_relu_slope", m.lrelu);
    getf("layer_norm_eps", m.ln_eps);
    geti("duration_predictor_kernel_size", m.dp_k);
    geti("depth_separable_num_layers", m.dds_layers);
    geti("duration_predictor_flow_bins", m.dp_bins);
    geti("duration_predictor_num_flows", m.dp_flows);
    {
        int tb = (int)m.dp_tail;  // ref parses with stoi (vits.cpp:861)
        geti("duration_predictor_tail_bound", tb);
        m.dp_tail = (float)tb;
    }
    getf("noise_scale_duration", m.noise_scale_dur);
    getf("noise_scale", m.noise_scale);
    getf("speaking_rate", m.speaking_rate);
    geti("sampling_rate", m.sampling_rate);
}

// ------------------------------------------------------------------------------------------------
// activations
// ------------------------------------------------------------------------------------------------
struct Act {
    int C = 0, T = 0;
    std::vector<float> d;
    Act() {}
    Act(int c, int t) : C(c), T(t), d((size_t)c * t, 0.f) {}
    float* row(int c) { return d.data() + (size_t)c * T; }
    const float* row(int c) const { return d.data() + (size_t)c * T; }
};

inline float leaky(float v, float slope) { return v > 0 ? v : v * slope; }  // ref: custom-ops.h:894-896
inline float sigmoidf(float v) { return 1.0f / (1.0f + std::exp(-v)); }     // ref: custom-ops.h:864-866
inline float softplusf(float x) {                                            // ref: custom-ops.h:872-879
    if (x > 20.f) return x;
    return (float)std::log(1.0 + std::exp((double)x));
}
inline float gelu_erf(float x) { return 0.5f * x * (1.0f + std::erf(x * 0.70710678118654752440f)); }  // HF:636

// ------------------------------------------------------------------------------------------------
// Q7: the reference's conv arithmetic. conv1d_impl unfolds the input with ggml_im2col_1d into an
// fp16 tensor and multiplies it with the fp16 weights (custom-ops.h:684-690; weights cast by scripts/export_vits.py:87), and
// ggml_conv_transpose_1d converts its source to fp16 the same way: every Conv1d / ConvTranspose1d (incl. the 192
// one-channel depthwise convs, vits.cpp:157-166) sees operands ROUNDED TO 16 BITS, products summed in fp32. Linear layers
// (q/k/v/out projections, ggml_mul_mat on f32 x f32, vits.cpp:287-289,358) do not. VO_ARITH_F16 / VO_ARITH_BF16 reproduce
// that (round-to-nearest-even of the conv input after its fused leaky_relu, and of the weights — a no-op when the file
// already stores that type); VO_ARITH_F32 (default) keeps everything fp32.
// ------------------------------------------------------------------------------------------------
thread_local int g_arith = 0;  // VO_ARITH_*; read on the calling thread at conv entry, captured by value in the workers

inline float round_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f;  // inf / nan
    u += 0x7fffu + ((u >> 16) & 1u);                 // round to nearest even on the upper 16 bits
    u &= 0xffff0000u;
    std::memcpy(&f, &u, 4);
    return f;
}
inline float round_f16(float f) {
    // fp32 -> fp16 (round to nearest even, subnormals kept, overflow to inf) -> fp32, in integer arithmetic
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    float out;
    if (a >= 0x7f800000u) return f;  // inf / nan
    if (a >= 0x477ff000u) {          // >= 65520: rounds to inf
        a = 0x7f800000u;
    } else if (a < 0x38800000u) {  // < 2^-14: fp16 subnormal, quantum 2^-24
        float v;
        std::memcpy(&v, &a, 4);
        const float q = v * 16777216.0f;                                  // exact scaling
        const float r = std::nearbyintf(q);                               // round-half-even in the default rounding mode
        v = r * (1.0f / 16777216.0f);
        std::memcpy(&a, &v, 4);
    } else {
        a += 0xfffu + ((a >> 13) & 1u);  // keep 10 mantissa bits
        a &= 0xffffe000u;
    }
    a |= sign;
    std::memcpy(&out, &a, 4);
    return out;
}
inline float round_arith(float v, int arith) { return arith == 2 ? round_f16(v) : arith == 1 ? round_bf16(v) : v; }

// ------------------------------------------------------------------------------------------------
// EMULATED ggml arithmetic (Q8) — INFERRED from upstream ggerganov/ggml of the reference's era (late 2023); the maxilevi/ggml fork the
// reference builds against is absent (empty submodule), so this is a labelled emulation, not a pinned restatement:
//   ggml_gelu (vits.cpp:673,687): GGML_GELU_FP16 is on by default — y = FP16_TO_FP32(table_gelu_f16[FP32_TO_FP16(x)]) with
//     table_gelu_f16[i] = FP32_TO_FP16(0.5 x (1 + tanhf(sqrt(2/pi) x (1 + 0.044715 x^2)))) of x = FP16_TO_FP32(i)      (ggml.c ggml_vec_gelu_f32)
//   ggml_soft_max (vits.cpp:329,719,735): val_i = FP16_TO_FP32(table_exp_f16[FP32_TO_FP16(x_i - max)]), table_exp_f16[i] = FP32_TO_FP16(expf(..)),
//     sum accumulated in double, then every val_i multiplied by (float)(1.0 / sum)                                  (ggml.c ggml_compute_forward_soft_max_f32)
// A table indexed by the fp16 bits is a pure function of the rounded argument, so round_f16(f(round_f16(x))) IS the table lookup.
// Selected per call with vo_opts.ggml_tables; default off (erf-GELU as in HF / fp32 soft-max).
// ------------------------------------------------------------------------------------------------
thread_local int g_ggml_tables = 0;
inline float gelu_tanh_f32(float x) { return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x))); }
inline float gelu_ggml_table(float x) { return round_f16(gelu_tanh_f32(round_f16(x))); }
inline float exp_ggml_table(float d) { return round_f16(expf(round_f16(d))); }
// in place over s[0..n), n > 0
inline void softmax_ggml_table(float* s, int n) {
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = std::max(mx, s[i]);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        if (s[i] == -INFINITY) s[i] = 0.f;
        else {
            s[i] = exp_ggml_table(s[i] - mx);
            sum += (double)s[i];
        }
    }
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < n; ++i) s[i] *= inv;
}
// weights rounded to the arithmetic type (cached per call site is not worth it: the oracle is a checker)
inline const float* rounded_weights(const float* w, size_t n, int arith, std::vector<float>& store) {
    if (!arith) return w;
    store.resize(n);
    for (size_t i = 0; i < n; ++i) store[i] = round_arith(w[i], arith);
    return store.data();
}

/*
 * conv1d_with_bias, ref: vits.cpp:171-176 -> conv1d_impl custom-ops.h:680-694 (im2col + mul_mat == direct conv),
 * bias broadcast over time custom-ops.h:397-431. Weights w[Cout][Cin][K] (torch layout, file ne=[K,Cin,Cout]).
 * Output length = T + pad_left + pad_right - (K-1)*dil. Optional fused leaky-relu on the input (the reference runs
 * it as a separate in-place node, vits.cpp:554,567,613,638).
 */
void conv1d_raw(const float* x, int cin, int T, int x_stride, const float* w, const float* bias, int cout, int K, int dil,
                int pad_l, int pad_r, bool pre_lrelu, float slope, float* y, int y_stride, int threads) {
    const int Tp = T + pad_l + pad_r;
    const int To = Tp - (K - 1) * dil;
    const int arith = g_arith;
    std::vector<float> wstore;
    w = rounded_weights(w, (size_t)cout * cin * K, arith, wstore);
    std::vector<float> xp((size_t)cin * Tp);
    parallel_for(threads, cin, [&](int64_t b0, int64_t e0) {
        for (int64_t ci = b0; ci < e0; ++ci) {
            float* row = xp.data() + (size_t)ci * Tp;
            std::fill(row, row + pad_l, 0.f);
            std::fill(row + pad_l + T, row + Tp, 0.f);
            float* dst = row + pad_l;
            const float* src = x + (size_t)ci * x_stride;
            if (pre_lrelu)
                for (int t = 0; t < T; ++t) dst[t] = leaky(src[t], slope);
            else
                std::memcpy(dst, src, sizeof(float) * T);
            if (arith)
                for (int t = 0; t < T; ++t) dst[t] = round_arith(dst[t], arith);  // the fp16 im2col (custom-ops.h:684-690)
        }
    });
    // register-blocked direct convolution: 4 output channels x 32 time steps of accumulators stay in vector registers
    // while (ci, tap) runs; each x vector is loaded once for the 4 channels. Same arithmetic as the im2col+GEMM of
    // the reference (sum over (ci, tap) of w*x, fp32), different summation grouping only.
    typedef float v8 __attribute__((vector_size(32), aligned(4)));
    const int TT = To <= 4096 ? 128 : 512, TB = 32;  // short sequences: smaller time tiles so that every thread gets work
    const int CB = 4;
    const int ncb = (cout + CB - 1) / CB;
    const int ntt = (To + TT - 1) / TT;
    parallel_for(threads, (int64_t)ncb * ntt, [&](int64_t b, int64_t e) {
        for (int64_t item = b; item < e; ++item) {
            const int cb = (int)(item / ntt), tt = (int)(item % ntt);
            const int co0 = cb * CB, nco = std::min(CB, cout - co0);
            const int t0 = tt * TT, nt = std::min(TT, To - t0);
            const float* wr[CB];
            float bv[CB];
            for (int r = 0; r < CB; ++r) {
                const int co = co0 + std::min(r, nco - 1);
                wr[r] = w + (size_t)co * cin * K;
                bv[r] = bias ? bias[co] : 0.f;
            }
            int tb = 0;
            for (; tb + TB <= nt; tb += TB) {
                v8 a[CB][TB / 8];
                for (int r = 0; r < CB; ++r)
                    for (int v = 0; v < TB / 8; ++v) a[r][v] = v8{bv[r], bv[r], bv[r], bv[r], bv[r], bv[r], bv[r], bv[r]};
                for (int ci = 0; ci < cin; ++ci) {
                    const float* xrow = xp.data() + (size_t)ci * Tp + t0 + tb;
                    for (int j = 0; j < K; ++j) {
                        const float* xr = xrow + j * dil;
                        v8 xv[TB / 8];
                        for (int v = 0; v < TB / 8; ++v) xv[v] = *reinterpret_cast<const v8*>(xr + 8 * v);
                        for (int r = 0; r < CB; ++r) {
                            const float wv = wr[r][(size_t)ci * K + j];
                            for (int v = 0; v < TB / 8; ++v) a[r][v] += wv * xv[v];
                        }
                    }
                }
                for (int r = 0; r < nco; ++r)
                    for (int v = 0; v < TB / 8; ++v) *reinterpret_cast<v8*>(y + (size_t)(co0 + r) * y_stride + t0 + tb + 8 * v) = a[r][v];
            }
            for (; tb < nt; ++tb) {  // scalar tail
                for (int r = 0; r < nco; ++r) {
                    float a = bv[r];
                    for (int ci = 0; ci < cin; ++ci)
                        for (int j = 0; j < K; ++j) a += wr[r][(size_t)ci * K + j] * xp[(size_t)ci * Tp + t0 + tb + j * dil];
                    y[(size_t)(co0 + r) * y_stride + t0 + tb] = a;
                }
            }
        }
    });
}

Act conv1d(const Act& x, const Tensor& w, const Tensor* b, int dil, int pad_l, int pad_r, bool pre_lrelu, float slope,
           int threads) {
    // file ne = [K, Cin, Cout]
    const int K = (int)w.ne[0], cin = (int)w.ne[1], cout = (int)w.ne[2];
    if (cin != x.C) throw std::runtime_error("conv1d: channel mismatch");
    const int To = x.T + pad_l + pad_r - (K - 1) * dil;
    Act y(cout, To);
    conv1d_raw(x.d.data(), cin, x.T, x.T, w.d.data(), b ? b->d.data() : nullptr, cout, K, dil, pad_l, pad_r, pre_lrelu, slope,
               y.d.data(), To, threads);
    return y;
}

/*
 * conv_transpose_1d_with_bias, ref: vits.cpp:178-193 -> ggml_conv_transpose_1d (p0 forced 0 at :187, Q1),
 * bias :190. HF: nn.ConvTranspose1d(padding=(k-s)/2) HF:483-490. Weights w[Cin][Cout][K] (torch layout, file
 * ne=[K,Cout,Cin]). y_full[co][i*s + k] += x[ci][i]*w[ci][co][k]; output = y_full[crop : len-crop].
 */
void conv_transpose1d_raw(const float* x, int cin, int T, int x_stride, const float* w, const float* bias, int cout, int K, int s,
                          int crop, float pre_slope, float* y, int y_stride, int threads = 0) {
    // y_full[co][i*s + k] += x[ci][i] * w[ci][co][k]  ==  per output phase r = n mod s (n = s*q + r):
    //   y_full[co][s*q + r] = sum_ci sum_m x[ci][q - m] * w[ci][co][r + s*m],  m < K/s   (same sum, regrouped so that the
    //   inner loop runs over q with unit stride). Output = y_full[crop : full - crop] + bias.
    const int full = (T - 1) * s + K;
    const int To = full - 2 * crop;
    const int taps = (K + s - 1) / s;        // taps per phase
    const int Q = T + taps - 1;              // q range: 0 .. T + taps - 2
    std::vector<float> xa((size_t)cin * (Q + taps), 0.f);  // x with (taps-1) zeros in front: xa[ci][q + taps-1 - m] = x[ci][q - m]
    const int XS = Q + taps;
    const int arith = g_arith;
    std::vector<float> wstore;
    w = rounded_weights(w, (size_t)cin * cout * K, arith, wstore);
    for (int ci = 0; ci < cin; ++ci)
        for (int t = 0; t < T; ++t) xa[(size_t)ci * XS + (taps - 1) + t] = round_arith(leaky(x[(size_t)ci * x_stride + t], pre_slope), arith);
    if (threads <= 0) threads = default_threads();
    parallel_for(threads, cout, [&](int64_t b, int64_t e) {
        std::vector<float> ph((size_t)s * Q);
        for (int64_t co = b; co < e; ++co) {
            std::fill(ph.begin(), ph.end(), 0.f);
            for (int ci = 0; ci < cin; ++ci) {
                const float* wk = w + ((size_t)ci * cout + co) * K;
                const float* xr = xa.data() + (size_t)ci * XS + (taps - 1);
                for (int r = 0; r < s; ++r) {
                    float* pr = ph.data() + (size_t)r * Q;
                    for (int mm = 0; mm < taps; ++mm) {
                        const int k = r + s * mm;
                        if (k >= K) break;
                        const float wv = wk[k];
                        const float* xs = xr - mm;  // x[q - mm]
                        for (int q = 0; q < Q; ++q) pr[q] += wv * xs[q];
                    }
                }
            }
            const float bv = bias ? bias[co] : 0.f;
            float* yo = y + (size_t)co * y_stride;
            for (int n = 0; n < To; ++n) {
                const int nf = n + crop;
                yo[n] = ph[(size_t)(nf % s) * Q + nf / s] + bv;
            }
        }
    });
}

/* layer_norm over channels, ref: vits.cpp:115-120 (ggml_norm_inplace + mul + add); HF nn.LayerNorm. */
void layer_norm_channels(Act& x, const Tensor& g, const Tensor& b, float eps) {
    const int C = x.C, T = x.T;
    for (int t = 0; t < T; ++t) {
        float mean = 0.f;
        for (int c = 0; c < C; ++c) mean += x.d[(size_t)c * T + t];
        mean /= C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) {
            float dv = x.d[(size_t)c * T + t] - mean;
            var += dv * dv;
        }
        var /= C;
        const float inv = 1.0f / std::sqrt(var + eps);
        for (int c = 0; c < C; ++c) {
            float& v = x.d[(size_t)c * T + t];
            v = (v - mean) * inv * g.d[c] + b.d[c];
        }
    }
}

/*
 * Relative-position self attention core, ref: vits.cpp:296-356 with helpers :195-235; HF:875-997.
 * q (already scaled, :296-297), k, v are [H*hd][T]; rel_k/rel_v [2w+1][hd] shared by heads (:323,348).
 * score_ij = q_i.k_j + [|j-i|<=w] q_i.Ek[j-i+w]; p = softmax_j; o_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij Ev[j-i+w].
 * (closed form of the pad/reshape skew trick)
 */
void rel_attention(const float* q, const float* k, const float* v, int H, int hd, int T, int stride, int len, int w, const float* Ek,
                   const float* Ev, float* out) {
    std::vector<float> s((size_t)len);
    for (int h = 0; h < H; ++h) {
        const float* qh = q + (size_t)h * hd * stride;
        const float* kh = k + (size_t)h * hd * stride;
        const float* vh = v + (size_t)h * hd * stride;
        float* oh = out + (size_t)h * hd * stride;
        for (int i = 0; i < len; ++i) {
            float mx = -INFINITY;
            for (int j = 0; j < len; ++j) {
                float a = 0.f;
                for (int d = 0; d < hd; ++d) a += qh[(size_t)d * stride + i] * kh[(size_t)d * stride + j];
                const int r = j - i + w;
                if (r >= 0 && r <= 2 * w) {
                    float bsum = 0.f;
                    for (int d = 0; d < hd; ++d) bsum += qh[(size_t)d * stride + i] * Ek[(size_t)r * hd + d];
                    a += bsum;
                }
                s[j] = a;
                mx = std::max(mx, a);
            }
            if (g_ggml_tables) {
                softmax_ggml_table(s.data(), len);  // ref :329 ggml_soft_max (emulated, Q8)
            } else {
                float sum = 0.f;
                for (int j = 0; j < len; ++j) {
                    s[j] = std::exp(s[j] - mx);
                    sum += s[j];
                }
                const float inv = 1.0f / sum;
                for (int j = 0; j < len; ++j) s[j] *= inv;
            }
            for (int d = 0; d < hd; ++d) {
                float a = 0.f;
                for (int j = 0; j < len; ++j) a += s[j] * vh[(size_t)d * stride + j];
                float bsum = 0.f;
                for (int r = 0; r <= 2 * w; ++r) {
                    const int j = i + r - w;
                    if (j >= 0 && j < len) bsum += s[j] * Ev[(size_t)r * hd + d];
                }
                oh[(size_t)d * stride + i] = a + bsum;
            }
        }
        for (int i = len; i < T; ++i)
            for (int d = 0; d < hd; ++d) oh[(size_t)d * stride + i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// reference noise stream, ref: vits.cpp:31 + ggml-util.h:187-199 (global default_random_engine, fresh
// normal_distribution<float> per tensor, filled in memory order)
// ------------------------------------------------------------------------------------------------
std::default_random_engine g_ref_rng;
std::mutex g_ref_rng_mu;
void ref_noise_fill(float* dst, size_t n) {
    std::lock_guard<std::mutex> lk(g_ref_rng_mu);
    std::normal_distribution<float> dist(0.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) dst[i] = dist(g_ref_rng);
}

// ------------------------------------------------------------------------------------------------
// the model graph
// ------------------------------------------------------------------------------------------------
struct Ctx {
    const vo_model& m;
    int mode;
    int threads;
    std::string prefix;
};

/* text encoder, ref: vits.cpp:244-440; HF VitsTextEncoder/VitsEncoderLayer HF:1051-1078, VitsFeedForward HF:1019-1039 */
void text_encoder(const Ctx& c, const int32_t* ids, int T, Act& enc_out, Act& m_p, Act& logs_p) {
    const vo_model& m = c.m;
    const int H = m.hidden;
    const Tensor& emb = m.T("text_encoder.embed_tokens.weight");  // [vocab][H]
    const int vocab = (int)emb.ne[1];
    Act x(H, T);
    const float sc = (float)std::sqrt((double)H);  // ref: vits.cpp:263
    for (int t = 0; t < T; ++t) {
        int id = ids[t];
        if (id < 0 || id >= vocab) throw std::runtime_error("token id out of range");
        for (int ch = 0; ch < H; ++ch) x.d[(size_t)ch * T + t] = emb.d[(size_t)id * H + ch] * sc;
    }
    const int hd = H / m.heads;
    const float scaling = (float)std::pow((double)hd, -0.5);  // ref: vits.cpp:296
    for (int l = 0; l < m.layers; ++l) {
        const std::string base = "text_encoder.encoder.layers." + std::to_string(l) + ".";
        auto lin = [&](const char* name) {
            const Tensor& W = m.T(base + "attention." + name + ".weight");  // [out][in]
            const Tensor& B = m.T(base + "attention." + name + ".bias");
            Act y(H, T);
            {
                const int keep = g_arith;
                g_arith = 0;  // Linear: ggml_mul_mat on f32 x f32 (vits.cpp:287-289,358), no fp16 im2col
                conv1d_raw(x.d.data(), H, T, T, W.d.data(), B.d.data(), H, 1, 1, 0, 0, false, 0.f, y.d.data(), T, c.threads);
                g_arith = keep;
            }
            return y;
        };
        Act q = lin("q_proj"), k = lin("k_proj"), v = lin("v_proj");
        for (auto& f : q.d) f *= scaling;
        Act att(H, T);
        rel_attention(q.d.data(), k.d.data(), v.d.data(), m.heads, hd, T, T, T, m.window, m.T(base + "attention.emb_rel_k").d.data(),
                      m.T(base + "attention.emb_rel_v").d.data(), att.d.data());
        Act o(H, T);
        {
            const Tensor& W = m.T(base + "attention.out_proj.weight");
            const Tensor& B = m.T(base + "attention.out_proj.bias");
            {
                const int keep = g_arith;
                g_arith = 0;  // Linear: ggml_mul_mat on f32 x f32 (vits.cpp:287-289,358), no fp16 im2col
                conv1d_raw(att.d.data(), H, T, T, W.d.data(), B.d.data(), H, 1, 1, 0, 0, false, 0.f, o.d.data(), T, c.threads);
                g_arith = keep;
            }
        }
        for (size_t i = 0; i < x.d.size(); ++i) x.d[i] = x.d[i] + o.d[i];  // ref: vits.cpp:367 (residual + cur)
        layer_norm_channels(x, m.T(base + "layer_norm.weight"), m.T(base + "layer_norm.bias"), m.ln_eps);
        // feed forward, ref: vits.cpp:377-407: pad (k-1)/2 left, k/2 right, conv_1, relu, pad, conv_2
        const int pl = (m.ffn_k - 1) / 2, pr = m.ffn_k / 2;
        Act h1 = conv1d(x, m.T(base + "feed_forward.conv_1.weight"), &m.T(base + "feed_forward.conv_1.bias"), 1, pl, pr, false, 0.f, c.threads);
        for (auto& f : h1.d) f = f > 0 ? f : 0.f;
        Act h2 = conv1d(h1, m.T(base + "feed_forward.conv_2.weight"), &m.T(base + "feed_forward.conv_2.bias"), 1, pl, pr, false, 0.f, c.threads);
        for (size_t i = 0; i < x.d.size(); ++i) x.d[i] = h2.d[i] + x.d[i];  // ref: vits.cpp:416 (cur + residual)
        layer_norm_channels(x, m.T(base + "final_layer_norm.weight"), m.T(base + "final_layer_norm.bias"), m.ln_eps);
    }
    enc_out = x;
    Act stats = conv1d(x, m.T("text_encoder.project.weight"), &m.T("text_encoder.project.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :429
    const int F = m.flow_size;
    m_p = Act(F, T);
    logs_p = Act(F, T);
    std::memcpy(m_p.d.data(), stats.d.data(), sizeof(float) * (size_t)F * T);  // ref: vits.cpp:436 split
    std::memcpy(logs_p.d.data(), stats.d.data() + (size_t)F * T, sizeof(float) * (size_t)F * T);
}

/* DDS conv, ref: vits.cpp:646-692 (depthwise conv :144-169 == grouped conv); HF:629-643 */
Act dds(const Ctx& c, const std::string& base, Act x, const Act* g) {
    const vo_model& m = c.m;
    const int C = x.C, T = x.T;
    if (g)
        for (size_t i = 0; i < x.d.size(); ++i) x.d[i] += g->d[i];  // ref :651-653
    for (int i = 0; i < m.dds_layers; ++i) {
        const Tensor& wd = m.T(base + "convs_dilated." + std::to_string(i) + ".weight");  // [C][1][K]
        const Tensor& bd = m.T(base + "convs_dilated." + std::to_string(i) + ".bias");
        const int K = m.dp_k;
        int dil = 1;
        for (int e = 0; e < i; ++e) dil *= K;  // ref :659 pow(kernel_size, i)
        const int pad = (K * dil - dil) / 2;   // ref :660
        Act h(C, T);
        for (int ch = 0; ch < C; ++ch) {
            for (int t = 0; t < T; ++t) {
                float a = bd.d[ch];
                for (int j = 0; j < K; ++j) {
                    const int tt = t + j * dil - pad;
                    if (tt >= 0 && tt < T) a += round_arith(wd.d[(size_t)ch * K + j], g_arith) * round_arith(x.d[(size_t)ch * T + tt], g_arith);
                }
                h.d[(size_t)ch * T + t] = a;
            }
        }
        layer_norm_channels(h, m.T(base + "norms_1." + std::to_string(i) + ".weight"), m.T(base + "norms_1." + std::to_string(i) + ".bias"), 1e-5f);
        for (auto& f : h.d) f = g_ggml_tables ? gelu_ggml_table(f) : gelu_erf(f);  // ref :673 ggml_gelu (tanh-approx fp16 table in ggml: Q8, emulated on request); HF erf
        Act p = conv1d(h, m.T(base + "convs_pointwise." + std::to_string(i) + ".weight"), &m.T(base + "convs_pointwise." + std::to_string(i) + ".bias"), 1,
                       0, 0, false, 0.f, c.threads);
        layer_norm_channels(p, m.T(base + "norms_2." + std::to_string(i) + ".weight"), m.T(base + "norms_2." + std::to_string(i) + ".bias"), 1e-5f);
        for (auto& f : p.d) f = g_ggml_tables ? gelu_ggml_table(f) : gelu_erf(f);  // ref :687
        for (size_t e = 0; e < x.d.size(); ++e) x.d[e] += p.d[e];  // ref :688
    }
    return x;
}

/*
 * Inverse rational-quadratic spline of one token's row, ref: vits.cpp:695-802 (rational_quadratic_spline); HF:211-302.
 * uw,uh: 10 unnormalised widths/heights (already / sqrt(filter_channels)); udp: the nb + 1 PADDED unnormalised derivatives as they
 * arrive at :704 (after the pad / index_put of :828-830 and, in reference mode, the masking of :837-840).
 * `q4` (reference mode, LAST token's row) selects the index -1 wrap (Q4, ggml-util.h:235-236,252-253): the writes
 * "[..., -1] = upper_bound" (:726,742) and "[..., -1] += 1e-6" (:750) never land on the last row (they land on the previous row's
 * last element, which is where the next row's write would have gone, so every other row is correct).
 */
static float spline_row(float x, const float* uw, const float* uh, const float* udp, int nb, float B, int mode, bool q4) {
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    std::vector<float> W(nb), Hh(nb), cw(nb + 1), chh(nb + 1), D(nb + 1);
    // widths
    {
        if (g_ggml_tables) {  // ref :719 ggml_soft_max (emulated, Q8)
            for (int i = 0; i < nb; ++i) W[i] = uw[i];
            softmax_ggml_table(W.data(), nb);
        } else {
            float mx = -INFINITY;
            for (int i = 0; i < nb; ++i) mx = std::max(mx, uw[i]);
            float sum = 0.f;
            for (int i = 0; i < nb; ++i) {
                W[i] = std::exp(uw[i] - mx);
                sum += W[i];
            }
            for (int i = 0; i < nb; ++i) W[i] /= sum;
        }
        if (mode == VO_MODE_REFERENCE) {
            const float sc = min_w + (1 - min_w * nb);  // ref :720 (Q3)
            for (int i = 0; i < nb; ++i) W[i] = W[i] * sc;
        } else {
            for (int i = 0; i < nb; ++i) W[i] = min_w + (1 - min_w * nb) * W[i];  // HF:225
        }
        float cum = 0.f;
        cw[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            cum += W[i];
            cw[i + 1] = cum;
        }
        for (int i = 0; i <= nb; ++i) cw[i] = (B - (-B)) * cw[i] + (-B);  // ref :724
        cw[0] = -B;                                                     // ref :725
        if (!q4) cw[nb] = B;                                            // ref :726
        for (int i = 0; i < nb; ++i) W[i] = cw[i + 1] - cw[i];          // ref :728-731
    }
    for (int i = 0; i <= nb; ++i) D[i] = min_d + softplusf(udp[i]);  // ref :733
    {
        if (g_ggml_tables) {  // ref :735
            for (int i = 0; i < nb; ++i) Hh[i] = uh[i];
            softmax_ggml_table(Hh.data(), nb);
        } else {
            float mx = -INFINITY;
            for (int i = 0; i < nb; ++i) mx = std::max(mx, uh[i]);
            float sum = 0.f;
            for (int i = 0; i < nb; ++i) {
                Hh[i] = std::exp(uh[i] - mx);
                sum += Hh[i];
            }
            for (int i = 0; i < nb; ++i) Hh[i] /= sum;
        }
        for (int i = 0; i < nb; ++i) Hh[i] = min_h + (1 - min_h * nb) * Hh[i];  // ref :736; HF:234
        float cum = 0.f;
        chh[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            cum += Hh[i];
            chh[i + 1] = cum;
        }
        for (int i = 0; i <= nb; ++i) chh[i] = (B - (-B)) * chh[i] + (-B);
        chh[0] = -B;
        if (!q4) chh[nb] = B;
        for (int i = 0; i < nb; ++i) Hh[i] = chh[i + 1] - chh[i];
    }
    // bin search on heights (reverse), ref :748-762; HF:243-245
    int bin = -1;
    for (int i = 0; i <= nb; ++i) {
        float loc = chh[i];
        if (i == nb && !q4) loc += 1e-6f;  // ref :750
        if (x >= loc) bin++;
    }
    bin = std::min(std::max(bin, 0), nb - 1);
    const float in_cw = cw[bin], in_w = W[bin], in_ch = chh[bin], in_h = Hh[bin];
    const float delta = Hh[bin] / W[bin];
    const float d0 = D[bin], d1 = D[bin + 1];
    const float i1 = d0 + d1 - 2 * delta;       // ref :775
    const float i2 = x - in_ch;                 // ref :782
    const float i3 = i2 * i1;                   // ref :783
    const float a = in_h * (delta - d0) + i3;   // ref :785
    const float b = in_h * d0 - i3;             // ref :786
    const float cc = -delta * i2;               // ref :787
    const float disc = b * b - 4 * a * cc;      // ref :789-791
    const float root = (2 * cc) / (-b - std::sqrt(disc));  // ref :792-795
    return root * in_w + in_cw;                 // ref :797
}

/* the padded derivative row of one token, ref :826-830: pad (1, 1), [..., 0] = constant, [..., -1] = constant — which misses the LAST
 * token's row in reference mode (Q4: the pad value 0 stays) */
static void padded_derivatives(const float* ud, int nb, int mode, bool last_token, float* udp) {
    const float min_d = 1e-3f;
    const float constant = (float)std::log(std::exp(1.0 - (double)min_d) - 1.0);  // ref :826
    udp[0] = constant;
    for (int i = 0; i < nb - 1; ++i) udp[i + 1] = ud[i];
    udp[nb] = (mode == VO_MODE_REFERENCE && last_token) ? 0.f : constant;
}

/*
 * HF semantics of the unconstrained spline for one token (HF:139-163): identity outside [-B, B], the spline inside. This is what the
 * reference's :804-852 computes too AS LONG AS every latent of the utterance lies inside the interval; when one does not, the
 * reference's masked get / set pair misaligns (Q6) — see unconstrained_spline_reference() below, which is what VO_MODE_REFERENCE runs.
 */
float spline_inverse(float x, const float* uw, const float* uh, const float* ud, int nb, float B, int mode, bool last_token) {
    if (!(x >= -B && x <= B)) return x;  // HF:143-151
    std::vector<float> udp(nb + 1);
    padded_derivatives(ud, nb, mode, last_token, udp.data());
    return spline_row(x, uw, uh, udp.data(), nb, B, mode, mode == VO_MODE_REFERENCE && last_token);
}

thread_local int64_t g_outside_latents = 0;  // latents outside [-B, B] met by the last duration_predictor call of this thread (all flows)

/*
 * unconstrained_rational_quadratic_spline, ref vits.cpp:804-852, LITERALLY (Q6), on one utterance. x [T]: the second half of the
 * latent; u [3nb-1][T]: conv_proj output. Statement by statement:
 *   :819-823  inside = (x >= -B) * (x <= B); outside = !inside
 *   :832      outputs = masked_set(zeros, outside, masked_get(x, INSIDE))  — masked_get KEEPS THE SHAPE (x where the mask is 1, else 0:
 *             custom-ops.h:746-749), masked_set consumes its values SEQUENTIALLY (values[index++] at every position whose mask is 1:
 *             custom-ops.h:836-850). So the j-th outside token receives element j of [x_t if inside_t else 0] — an unrelated token's
 *             latent or 0 —, not its own latent (HF: identity tails).
 *   :834-835  the spline input of EVERY token is masked_get(x, inside): outside tokens enter as 0 ...
 *   :837-840  ... with their unnormalised widths / heights / padded derivatives (constants of :829-830 included) zeroed by the same mask
 *   :849      outputs = masked_set(outputs, inside, result): the k-th INSIDE token receives result[k] — the spline output of token k,
 *             which is its own only while no token before it lay outside. One outside latent shifts every later token's value.
 * With every latent inside, all of this is the identity permutation and equals spline_inverse() per token. (The reference's own
 * known-answer test for masked_get expects the COMPACTED form {2,4,5} and is commented out: test/test_ggml_utils.cpp:584-590.)
 */
static void unconstrained_spline_reference(float* x, const float* u, int T, int nb, float B, float inv_sqrt) {
    std::vector<float> inside(T), vals(T), res(T), out(T, 0.f), uw(nb), uh(nb), ud(nb - 1), udp(nb + 1);
    for (int t = 0; t < T; ++t) {
        inside[t] = (x[t] >= -B && x[t] <= B) ? 1.f : 0.f;  // :819-823
        vals[t] = inside[t] == 1.f ? x[t] : 0.f;            // masked_get(inputs, inside_interval_mask), shape kept
        if (inside[t] != 1.f) ++g_outside_latents;
    }
    {
        int index = 0;  // :832 masked_set(outputs, outside_interval_mask, vals)
        for (int t = 0; t < T; ++t)
            if (inside[t] != 1.f) out[t] = vals[index++];
    }
    for (int t = 0; t < T; ++t) {
        const float mk = inside[t];
        for (int i = 0; i < nb; ++i) uw[i] = mk == 1.f ? u[(size_t)i * T + t] * inv_sqrt : 0.f;         // :878-880, :838
        for (int i = 0; i < nb; ++i) uh[i] = mk == 1.f ? u[(size_t)(nb + i) * T + t] * inv_sqrt : 0.f;  // :881-883, :839
        for (int i = 0; i < nb - 1; ++i) ud[i] = u[(size_t)(2 * nb + i) * T + t];                        // :885
        padded_derivatives(ud.data(), nb, VO_MODE_REFERENCE, t == T - 1, udp.data());                    // :828-830
        if (mk != 1.f)
            for (int i = 0; i <= nb; ++i) udp[i] = 0.f;                                                  // :840
        res[t] = spline_row(vals[t], uw.data(), uh.data(), udp.data(), nb, B, VO_MODE_REFERENCE, t == T - 1);
    }
    {
        int index = 0;  // :849 masked_set(outputs, inside_interval_mask, result)
        for (int t = 0; t < T; ++t)
            if (inside[t] == 1.f) out[t] = res[index++];
    }
    std::memcpy(x, out.data(), sizeof(float) * T);
}

/* stochastic duration predictor (reverse), ref: vits.cpp:927-972, conv flow :855-899, affine :901-925; HF:740-804 */
Act duration_predictor(const Ctx& c, const Act& enc_out, const float* noise /*[2][T]*/) {
    const vo_model& m = c.m;
    const int T = enc_out.T;
    const std::string dp = "duration_predictor.";
    Act x = conv1d(enc_out, m.T(dp + "conv_pre.weight"), &m.T(dp + "conv_pre.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :934
    x = dds(c, dp + "conv_dds.", x, nullptr);                                                                           // ref :941
    Act cond = conv1d(x, m.T(dp + "conv_proj.weight"), &m.T(dp + "conv_proj.bias"), 1, 0, 0, false, 0.f, c.threads);    // ref :943
    Act z(2, T);
    g_outside_latents = 0;
    for (int i = 0; i < 2 * T; ++i) z.d[i] = noise[i] * m.noise_scale_dur;  // ref :948-949
    const int nb = m.dp_bins;
    const float inv_sqrt = (float)(1.0 / std::sqrt((double)m.hidden));  // ref :877
    for (int f = m.dp_flows; f > -1; --f) {  // ref :953-965 (flow 1 skipped: HF "remove a useless vflow" HF:792)
        if (f == 1) continue;
        // flip channels, ref :956
        for (int t = 0; t < T; ++t) std::swap(z.d[t], z.d[(size_t)T + t]);
        const std::string fb = dp + "flows." + std::to_string(f) + ".";
        if (f == 0) {
            // elementwise affine reverse, ref :901-925 (Q5: exp(+log_scale)); HF:703 exp(-log_scale)
            const Tensor& tr = m.T(fb + "translate");
            const Tensor& ls = m.T(fb + "log_scale");
            for (int ch = 0; ch < 2; ++ch) {
                const float e = std::exp(c.mode == VO_MODE_REFERENCE ? ls.d[ch] : -ls.d[ch]);
                for (int t = 0; t < T; ++t) z.d[(size_t)ch * T + t] = (z.d[(size_t)ch * T + t] - tr.d[ch]) * e;
            }
        } else {
            Act z0(1, T);
            std::memcpy(z0.d.data(), z.d.data(), sizeof(float) * T);
            Act h = conv1d(z0, m.T(fb + "conv_pre.weight"), &m.T(fb + "conv_pre.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :864
            h = dds(c, fb + "conv_dds.", h, &cond);                                                                       // ref :868
            Act u = conv1d(h, m.T(fb + "conv_proj.weight"), &m.T(fb + "conv_proj.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :871, [3nb-1][T]
            if (c.mode == VO_MODE_REFERENCE) {
                // the reference's masked get / set pair, literally (Q6): equals the per-token form below while every latent is inside
                unconstrained_spline_reference(z.d.data() + T, u.d.data(), T, nb, m.dp_tail, inv_sqrt);
            } else {
                std::vector<float> uw(nb), uh(nb), ud(nb - 1);
                for (int t = 0; t < T; ++t) {
                    for (int i = 0; i < nb; ++i) uw[i] = u.d[(size_t)i * T + t] * inv_sqrt;         // ref :878-880
                    for (int i = 0; i < nb; ++i) uh[i] = u.d[(size_t)(nb + i) * T + t] * inv_sqrt;  // ref :881-883
                    for (int i = 0; i < nb - 1; ++i) ud[i] = u.d[(size_t)(2 * nb + i) * T + t];     // ref :885
                    const float xin = z.d[(size_t)T + t];
                    if (!(xin >= -m.dp_tail && xin <= m.dp_tail)) ++g_outside_latents;
                    z.d[(size_t)T + t] = spline_inverse(xin, uw.data(), uh.data(), ud.data(), nb, m.dp_tail, c.mode, t == T - 1);
                }
            }
        }
    }
    Act logw(1, T);
    std::memcpy(logw.d.data(), z.d.data(), sizeof(float) * T);  // ref :967-968
    return logw;
}

/* WaveNet, ref: vits.cpp:452-498 (gate :442-450); HF:347-374 */
Act wavenet(const Ctx& c, const std::string& base, Act h) {
    const vo_model& m = c.m;
    const int Hc = m.hidden, T = h.T;
    Act out(Hc, T);
    for (int l = 0; l < m.wn_layers; ++l) {
        int dil = 1;
        for (int e = 0; e < l; ++e) dil *= m.wn_rate;                  // ref :469
        const int pad = (m.wn_k * dil - dil) / 2;                     // ref :470
        Act a = conv1d(h, m.T(base + "in_layers." + std::to_string(l) + ".weight"), &m.T(base + "in_layers." + std::to_string(l) + ".bias"), dil, pad, pad,
                       false, 0.f, c.threads);
        Act g(Hc, T);
        for (int ch = 0; ch < Hc; ++ch)
            for (int t = 0; t < T; ++t) g.d[(size_t)ch * T + t] = std::tanh(a.d[(size_t)ch * T + t]) * sigmoidf(a.d[(size_t)(ch + Hc) * T + t]);
        Act rs = conv1d(g, m.T(base + "res_skip_layers." + std::to_string(l) + ".weight"), &m.T(base + "res_skip_layers." + std::to_string(l) + ".bias"), 1, 0,
                        0, false, 0.f, c.threads);
        if (l < m.wn_layers - 1) {  // ref :484-489
            for (size_t i = 0; i < (size_t)Hc * T; ++i) h.d[i] += rs.d[i];
            for (size_t i = 0; i < (size_t)Hc * T; ++i) out.d[i] += rs.d[(size_t)Hc * T + i];
        } else {
            for (size_t i = 0; i < (size_t)Hc * T; ++i) out.d[i] += rs.d[i];  // ref :491
        }
    }
    return out;
}

/* residual coupling flow, reverse, ref: vits.cpp:519-538,500-517; HF:588-597,563-578 */
Act flow_reverse(const Ctx& c, Act x) {
    const vo_model& m = c.m;
    const int F = m.flow_size, half = F / 2, T = x.T;
    for (int i = m.n_flows - 1; i > -1; --i) {
        // flip channels, ref :532 (custom-ops.h:218-245)
        for (int ch = 0; ch < F / 2; ++ch)
            for (int t = 0; t < T; ++t) std::swap(x.d[(size_t)ch * T + t], x.d[(size_t)(F - 1 - ch) * T + t]);
        const std::string fb = "flow.flows." + std::to_string(i) + ".";
        Act x0(half, T);
        std::memcpy(x0.d.data(), x.d.data(), sizeof(float) * (size_t)half * T);  // ref :502
        Act h = conv1d(x0, m.T(fb + "conv_pre.weight"), &m.T(fb + "conv_pre.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :503
        Act o = wavenet(c, fb + "wavenet.", h);                                                                       // ref :505
        Act mean = conv1d(o, m.T(fb + "conv_post.weight"), &m.T(fb + "conv_post.bias"), 1, 0, 0, false, 0.f, c.threads);  // ref :506
        for (size_t e = 0; e < (size_t)half * T; ++e) x.d[(size_t)half * T + e] -= mean.d[e];  // ref :513
    }
    return x;
}

/* HiFiGAN, ref: vits.cpp:583-644, resblock :545-581, conv transpose :178-193; HF:519-551,455-463 */
struct StageTimer {  // VO_TIMING=1 prints where the CPU time goes (developer aid for the cpu_baseline leg)
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() {
        static const bool on = std::getenv("VO_TIMING") != nullptr;
        if (on) std::fprintf(stderr, "[oracle] %-18s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

void hifigan(const Ctx& c, const Act& z, Act& pre_tanh, Act& wave) {
    const vo_model& m = c.m;
    const float slope = m.lrelu;
    Act h = conv1d(z, m.T("decoder.conv_pre.weight"), &m.T("decoder.conv_pre.bias"), 1, 3, 3, false, 0.f, c.threads);  // ref :601
    const int nk = (int)m.rb_k.size();
    for (size_t i = 0; i < m.up_rates.size(); ++i) {
        const Tensor& w = m.T("decoder.upsampler." + std::to_string(i) + ".weight");  // file ne=[K,Cout,Cin]
        const Tensor& b = m.T("decoder.upsampler." + std::to_string(i) + ".bias");
        const int K = (int)w.ne[0], cout = (int)w.ne[1], cin = (int)w.ne[2], s = m.up_rates[i];
        const int crop = (c.mode == VO_MODE_REFERENCE) ? 0 : (m.up_k[i] - s) / 2;  // ref :187 (Q1) / HF:488
        const int To = (h.T - 1) * s + K - 2 * crop;
        Act u(cout, To);
        StageTimer* tt = new StageTimer("convT");
        conv_transpose1d_raw(h.d.data(), cin, h.T, h.T, w.d.data(), b.d.data(), cout, K, s, crop, slope, u.d.data(), To);  // lrelu ref :613
        delete tt;
        StageTimer trb("resblocks(stage)");
        Act sum;
        for (int j = 0; j < nk; ++j) {
            const int idx = (int)i * nk + j;
            const std::string rb = "decoder.resblocks." + std::to_string(idx) + ".";
            const int k = m.rb_k[j];
            Act y = u;  // residual
            for (size_t di = 0; di < m.rb_d[j].size(); ++di) {
                const int d = m.rb_d[j][di];
                const int p1 = (k * d - d) / 2, p2 = (k - 1) / 2;  // ref :541-543
                Act t1 = conv1d(y, m.T(rb + "convs1." + std::to_string(di) + ".weight"), &m.T(rb + "convs1." + std::to_string(di) + ".bias"), d, p1, p1, true,
                                slope, c.threads);
                Act t2 = conv1d(t1, m.T(rb + "convs2." + std::to_string(di) + ".weight"), &m.T(rb + "convs2." + std::to_string(di) + ".bias"), 1, p2, p2, true,
                                slope, c.threads);
                parallel_for(c.threads, y.C, [&](int64_t b0, int64_t e0) {
                    for (size_t e = (size_t)b0 * y.T; e < (size_t)e0 * y.T; ++e) y.d[e] = y.d[e] + t2.d[e];  // ref :578
                });
            }
            if (j == 0) sum = y;
            else
                parallel_for(c.threads, sum.C, [&](int64_t b0, int64_t e0) {
                    for (size_t e = (size_t)b0 * sum.T; e < (size_t)e0 * sum.T; ++e) sum.d[e] += y.d[e];  // ref :630
                });
        }
        if (c.mode == VO_MODE_REFERENCE) {
            const float sc = (float)(1.0 / nk);  // ref :607,635
            for (auto& f : sum.d) f *= sc;
        } else {
            for (auto& f : sum.d) f /= (float)nk;  // HF:546
        }
        h = sum;
    }
    const float final_slope = (c.mode == VO_MODE_REFERENCE) ? slope : 0.01f;  // ref :638 (Q2) / HF:548
    pre_tanh = conv1d(h, m.T("decoder.conv_post.weight"), nullptr, 1, 3, 3, true, final_slope, c.threads);  // ref :639
    wave = pre_tanh;
    for (auto& f : wave.d) f = std::tanh(f);  // ref :642
}

}  // namespace

struct vo_run {
    std::map<std::string, Act> taps;
};

// ------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------
VO_API const char* vo_last_error(void) { return g_err.c_str(); }

VO_API vo_model* vo_load(const char* bytes, size_t size) {
    try {
        Reader r{(const uint8_t*)bytes, size};
        auto m = std::make_unique<vo_model>();
        uint32_t nv = r.u32();
        for (uint32_t i = 0; i < nv && r.ok; ++i) {
            std::string k = r.str();
            uint32_t id = r.u32();
            m->vocab[k] = (int32_t)id;
        }
        m->add_blank = r.u32();
        m->normalize = r.u32();
        m->pad_token = r.str();
        m->unk_token = r.str();
        uint32_t nc = r.u32();
        for (uint32_t i = 0; i < nc && r.ok; ++i) {
            std::string k = r.str();
            std::string v = r.str();
            m->config[k] = v;
        }
        uint32_t nt = r.u32();
        for (uint32_t i = 0; i < nt && r.ok; ++i) {
            std::string name = r.str();
            Tensor t;
            t.dtype = (int)r.u32();
            t.rank = (int)r.u32();
            if (t.rank > 4) throw std::runtime_error("rank > 4");
            int64_t n = 1;
            for (int j = 0; j < t.rank; ++j) {
                t.ne[j] = r.u32();
                n *= t.ne[j];
            }
            uint32_t nbytes = r.u32();
            if (!r.ok || r.off + nbytes > r.n) throw std::runtime_error("truncated tensor " + name);
            t.d.resize((size_t)n);
            if (t.dtype == 0) {
                if (nbytes != n * 4) throw std::runtime_error("bad f32 size " + name);
                std::memcpy(t.d.data(), r.p + r.off, nbytes);
            } else if (t.dtype == 1 || t.dtype == 2) {  // 2 = bf16, this repo's extension of the format
                if (nbytes != n * 2) throw std::runtime_error("bad 16-bit size " + name);
                for (int64_t e = 0; e < n; ++e) {
                    uint16_t hv;
                    std::memcpy(&hv, r.p + r.off + 2 * e, 2);
                    if (t.dtype == 1) t.d[(size_t)e] = half_to_float(hv);
                    else {
                        uint32_t bits = (uint32_t)hv << 16;
                        std::memcpy(&t.d[(size_t)e], &bits, 4);
                    }
                }
            } else
                throw std::runtime_error("Unsupported tensor type");  // ref: vits_model_data.cpp:85
            r.off += nbytes;
            m->tensors[name] = std::move(t);
        }
        if (!r.ok) throw std::runtime_error("truncated model file");
        load_hparams(*m);
        return m.release();
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

VO_API void vo_free(vo_model* m) { delete m; }
VO_API int32_t vo_num_tensors(const vo_model* m) { return (int32_t)m->tensors.size(); }

VO_API int64_t vo_tensor(const vo_model* m, const char* name, float* dst, size_t cap, int32_t* dtype, int32_t* rank, int64_t* dims4) {
    auto it = m->tensors.find(name);
    if (it == m->tensors.end()) return -1;
    const Tensor& t = it->second;
    if (dtype) *dtype = t.dtype;
    if (rank) *rank = t.rank;
    if (dims4)
        for (int i = 0; i < 4; ++i) dims4[i] = t.ne[i];
    if (dst) std::memcpy(dst, t.d.data(), sizeof(float) * std::min(cap, t.d.size()));
    return t.n();
}

VO_API int64_t vo_config(const vo_model* m, const char* key, char* dst, size_t cap) {
    auto it = m->config.find(key);
    if (it == m->config.end()) return -1;
    if (dst && cap) {
        size_t n = std::min(cap - 1, it->second.size());
        std::memcpy(dst, it->second.data(), n);
        dst[n] = 0;
    }
    return (int64_t)it->second.size();
}

// ggml_tables == 1: stage one in the exact order of include/vits_exact_math.h (vits_oracle_exact.cpp), shared with the product's device code so that
// the durations of that mode are bit-identical on both sides; == 2: the loops of THIS file with the table lookups (the independent restatement the
// exact one is compared against at tolerance). Requires fp32 stage-one arithmetic (default scope).
static bool exact_stage_one_wanted(const vo_opts* opts) { return opts && opts->ggml_tables == 1; }
static void run_exact_stage_one(const vo_model& m, const vo_opts* opts, const int32_t* ids, int T, const float* noise, int threads, vo_exact::StageOne& out) {
    if (g_arith) throw std::runtime_error("ggml_tables = 1 (exact order) needs fp32 stage-one arithmetic: use VO_SCOPE_FLOW_VOCODER");
    vo_exact::ModelView v;
    v.T = [&m](const std::string& name) {
        const Tensor& t = m.T(name);
        vo_exact::TensorRef r;
        r.d = t.d.data();
        r.rank = t.rank;
        for (int i = 0; i < 4; ++i) r.ne[i] = t.ne[i];
        return r;
    };
    v.hidden = m.hidden, v.layers = m.layers, v.heads = m.heads, v.window = m.window, v.ffn_k = m.ffn_k, v.flow_size = m.flow_size;
    v.dp_k = m.dp_k, v.dds_layers = m.dds_layers, v.dp_bins = m.dp_bins, v.dp_flows = m.dp_flows;
    v.ln_eps = m.ln_eps, v.dp_tail = m.dp_tail, v.noise_scale_dur = m.noise_scale_dur, v.speaking_rate = m.speaking_rate;
    vo_exact::stage_one(v, (opts ? opts->mode : VO_MODE_REFERENCE) == VO_MODE_REFERENCE, ids, T, noise, threads, out);
    g_outside_latents = out.outside_latents;
}

VO_API vo_run* vo_process_ids(vo_model* mp, const int32_t* ids, int32_t T, const vo_opts* opts) {
    try {
        const vo_model& m = *mp;
        if (T <= 0) throw std::runtime_error("empty input");
        const int arith_all = opts ? opts->arith : 0;
        // scope FLOW_VOCODER: the text encoder and the duration predictor run in exact fp32, the 16-bit operand rounding starts at the flow
        g_arith = (opts && opts->arith_scope == VO_SCOPE_ALL_CONVS) ? arith_all : 0;
        g_ggml_tables = opts ? opts->ggml_tables : 0;
        struct ArithReset {
            ~ArithReset() { g_arith = 0, g_ggml_tables = 0; }
        } arith_reset;
        Ctx c{m, opts ? opts->mode : VO_MODE_REFERENCE, (opts && opts->threads > 0) ? opts->threads : default_threads(), ""};
        auto run = std::make_unique<vo_run>();
        Act enc, m_p, logs_p;
        const bool exact = exact_stage_one_wanted(opts);
        if (!exact) {
            StageTimer t("text_encoder");
            text_encoder(c, ids, T, enc, m_p, logs_p);
        }
        // duration noise [2][T], ref: vits.cpp:948 tensor_randn{T,2,1} (memory order: channel-major, time fastest)
        Act nd(2, T);
        const int nk = opts ? opts->noise_kind : VO_NOISE_REFERENCE;
        if (nk == VO_NOISE_EXPLICIT) std::memcpy(nd.d.data(), opts->noise_dur, sizeof(float) * 2 * T);
        else if (nk == VO_NOISE_COUNTER)
            for (int i = 0; i < 2 * T; ++i) nd.d[i] = vits_counter_normal(opts->noise_seed, VITS_STREAM_NOISE_DUR, (uint64_t)i);
        else
            ref_noise_fill(nd.d.data(), (size_t)2 * T);
        Act logw;
        vo_exact::StageOne ex;
        if (exact) {
            StageTimer t("stage_one_exact");
            run_exact_stage_one(m, opts, ids, T, nd.d.data(), c.threads, ex);
            const int F = m.flow_size;
            enc = Act(m.hidden, T), m_p = Act(F, T), logs_p = Act(F, T), logw = Act(1, T);
            enc.d = ex.enc;
            std::memcpy(m_p.d.data(), ex.stats.data(), sizeof(float) * (size_t)F * T);
            std::memcpy(logs_p.d.data(), ex.stats.data() + (size_t)F * T, sizeof(float) * (size_t)F * T);
            logw.d = ex.logw;
        } else {
            StageTimer t("duration_predictor");
            logw = duration_predictor(c, enc, nd.d.data());
        }
        // durations, ref: vits.cpp:995-1001 ; HF:1348-1349
        Act dur(1, T);
        const float length_scale = (float)(1.0 / m.speaking_rate);
        double total = 0;
        for (int t = 0; t < T; ++t) {
            float d = exact ? ex.dur[t] : std::ceil(std::exp(logw.d[t]) * length_scale);
            if (opts && opts->fixed_duration > 0) d = (float)opts->fixed_duration;
            dur.d[t] = d;
            total += d;
        }
        const int L = (int)std::max(1.0, total);  // ref :999 clamp(.,1) then (int) at :1133
        // monotonic alignment as a gather, ref: vits.cpp:1028-1057
        std::vector<int> a((size_t)L, -1);
        {
            double cum = 0;
            int j = 0;
            for (int t = 0; t < T; ++t) {
                cum += dur.d[t];
                for (; j < L && j < cum; ++j) a[j] = t;
            }
        }
        const int F = m.flow_size;
        Act np(F, L);
        if (nk == VO_NOISE_EXPLICIT) {
            for (int ch = 0; ch < F; ++ch) std::memcpy(np.row(ch), opts->noise_prior + (size_t)ch * opts->noise_prior_stride, sizeof(float) * L);
        } else if (nk == VO_NOISE_COUNTER) {
            for (int ch = 0; ch < F; ++ch)
                for (int t = 0; t < L; ++t) np.d[(size_t)ch * L + t] = vits_counter_normal(opts->noise_seed, VITS_STREAM_NOISE_PRIOR, (uint64_t)ch * L + t);
        } else {
            ref_noise_fill(np.d.data(), (size_t)F * L);  // ref :1059 tensor_randn_like(prior_means ne=[L,192])
        }
        Act z_p(F, L);
        for (int ch = 0; ch < F; ++ch)
            for (int j = 0; j < L; ++j) {
                const int t = a[j];
                const float mu = t >= 0 ? m_p.d[(size_t)ch * T + t] : 0.f;
                const float ls = t >= 0 ? logs_p.d[(size_t)ch * T + t] : 0.f;
                float n = np.d[(size_t)ch * L + j] * std::exp(ls);  // ref :1060
                n = n * m.noise_scale;                                // ref :1061
                z_p.d[(size_t)ch * L + j] = mu + n;                   // ref :1063
            }
        g_arith = arith_all;
        Act z;
        {
            StageTimer t("flow");
            z = flow_reverse(c, z_p);
        }
        Act pre, wave;
        {
            StageTimer t("hifigan");
            hifigan(c, z, pre, wave);
        }
        run->taps["enc_out"] = enc;
        run->taps["prior_mean"] = m_p;
        run->taps["prior_logvar"] = logs_p;
        run->taps["log_duration"] = logw;
        run->taps["durations"] = dur;
        run->taps["noise_dur"] = nd;
        run->taps["noise_prior"] = np;
        run->taps["z_p"] = z_p;
        run->taps["z_flow"] = z;
        run->taps["pre_tanh"] = pre;
        run->taps["waveform"] = wave;
        return run.release();
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

/* stage one only (text encoder + duration predictor, ~1 % of the work): the log-durations of one utterance and the durations
 * ceil(exp(logw) * length_scale) (vits.cpp:995-1001). Used by bench.py's duration-boundary report, which needs all 8,192 ids of the
 * benchmark batch and cannot afford 64 full vocoder runs. Same code path as vo_process_ids up to that point. */
VO_API int vo_log_durations(vo_model* mp, const int32_t* ids, int32_t T, const vo_opts* opts, float* logw_out, float* dur_out) {
    try {
        const vo_model& m = *mp;
        if (T <= 0) throw std::runtime_error("empty input");
        g_arith = (opts && opts->arith_scope == VO_SCOPE_ALL_CONVS) ? opts->arith : 0;
        g_ggml_tables = opts ? opts->ggml_tables : 0;
        struct ArithReset {
            ~ArithReset() { g_arith = 0, g_ggml_tables = 0; }
        } arith_reset;
        Ctx c{m, opts ? opts->mode : VO_MODE_REFERENCE, (opts && opts->threads > 0) ? opts->threads : default_threads(), ""};
        Act enc, m_p, logs_p;
        const bool exact = exact_stage_one_wanted(opts);
        if (!exact) text_encoder(c, ids, T, enc, m_p, logs_p);
        Act nd(2, T);
        const int nk = opts ? opts->noise_kind : VO_NOISE_REFERENCE;
        if (nk == VO_NOISE_EXPLICIT) std::memcpy(nd.d.data(), opts->noise_dur, sizeof(float) * 2 * T);
        else if (nk == VO_NOISE_COUNTER)
            for (int i = 0; i < 2 * T; ++i) nd.d[i] = vits_counter_normal(opts->noise_seed, VITS_STREAM_NOISE_DUR, (uint64_t)i);
        else
            ref_noise_fill(nd.d.data(), (size_t)2 * T);
        if (exact) {
            vo_exact::StageOne ex;
            run_exact_stage_one(m, opts, ids, T, nd.d.data(), c.threads, ex);
            for (int t = 0; t < T; ++t) {
                if (logw_out) logw_out[t] = ex.logw[t];
                if (dur_out) dur_out[t] = ex.dur[t];
            }
            return 0;
        }
        Act logw = duration_predictor(c, enc, nd.d.data());
        const float length_scale = (float)(1.0 / m.speaking_rate);
        for (int t = 0; t < T; ++t) {
            if (logw_out) logw_out[t] = logw.d[t];
            if (dur_out) dur_out[t] = std::ceil(std::exp(logw.d[t]) * length_scale);
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

VO_API int64_t vo_run_tap(const vo_run* r, const char* name, float* dst, size_t cap) {
    auto it = r->taps.find(name);
    if (it == r->taps.end()) return 0;
    if (dst) std::memcpy(dst, it->second.d.data(), sizeof(float) * std::min(cap, it->second.d.size()));
    return (int64_t)it->second.d.size();
}
VO_API void vo_run_free(vo_run* r) { delete r; }

/* tokenizer, ref: src/vits_tokenizer.cpp:57-78 (greedy vocab match) and :182-208 (lower-case, blanks).
 * The reference iterates an unordered_map (order unspecified, Q11); this restatement takes the longest match,
 * which is identical for prefix-free vocabularies such as the single-character MMS ones. */
VO_API int64_t vo_tokenize(const vo_model* m, const char* text, int32_t* ids, size_t cap) {
    std::string s(text);
    for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
    std::vector<int32_t> toks;
    size_t i = 0;
    while (i < s.size()) {
        size_t best = 0;
        int32_t best_id = 0;
        for (auto& kv : m->vocab) {
            const std::string& k = kv.first;
            if (!k.empty() && k.size() > best && s.compare(i, k.size(), k) == 0) {
                best = k.size();
                best_id = kv.second;
            }
        }
        if (best) {
            toks.push_back(best_id);
            i += best;
        } else
            i++;
    }
    std::vector<int32_t> fin;
    if (m->add_blank) {  // ref :200-206 ; returns empty when add_blank is false (:199-207)
        auto it = m->vocab.find(m->pad_token);
        int32_t blank = it != m->vocab.end() ? it->second : 0;
        fin.assign(toks.size() * 2 + 1, blank);
        for (size_t k = 0; k < toks.size(); ++k) fin[k * 2 + 1] = toks[k];
    }
    for (size_t k = 0; k < fin.size() && k < cap; ++k) ids[k] = fin[k];
    return (int64_t)fin.size();
}

VO_API void vo_reference_noise_seed(uint32_t seed) {
    std::lock_guard<std::mutex> lk(g_ref_rng_mu);
    g_ref_rng.seed(seed);
}
VO_API void vo_reference_noise_draw(float* dst, size_t n) { ref_noise_fill(dst, n); }

// ---- operator-level -----------------------------------------------------------------------------
VO_API int vo_conv1d(const vo_conv1d_desc* d, const float* x, const float* w, const float* bias, const float* residual, const float* accum,
                     const int32_t* lens, float* y, int32_t threads) {
    const int gate = d->post_act == 2;
    const int cy = gate ? d->cout / 2 : d->cout;
    g_arith = d->arith;
    struct ArithReset {
        ~ArithReset() { g_arith = 0; }
    } arith_reset;
    const int pad_r = (d->k - 1) * d->dilation - d->pad_left;
    if (pad_r < 0) return -1;
    if (threads <= 0) threads = default_threads();
    for (int b = 0; b < d->batch; ++b) {
        const int len = lens ? lens[b] : d->t;
        std::vector<float> tmp((size_t)d->cout * len);
        conv1d_raw(x + (size_t)b * d->cin * d->t_stride, d->cin, len, d->t_stride, w, bias, d->cout, d->k, d->dilation, d->pad_left, pad_r, d->pre_act == 1,
                   d->pre_slope, tmp.data(), len, threads);
        float* yb = y + (size_t)b * cy * d->t_stride;
        for (int c = 0; c < cy; ++c)
            for (int t = 0; t < len; ++t) {
                float v;
                if (gate) v = std::tanh(tmp[(size_t)c * len + t]) * sigmoidf(tmp[(size_t)(c + cy) * len + t]);
                else {
                    v = tmp[(size_t)c * len + t];
                    if (d->post_act == 1) v = v > 0 ? v : 0.f;
                }
                const size_t o = (size_t)b * cy * d->t_stride + (size_t)c * d->t_stride + t;
                if (residual) v = residual[o] + v;
                if (accum) v = (accum[o] + v) * d->out_scale;
                yb[(size_t)c * d->t_stride + t] = v;
            }
    }
    return 0;
}

VO_API int vo_conv_transpose1d(const vo_convt1d_desc* d, const float* x, const float* w, const float* bias, const int32_t* lens, float* y) {
    g_arith = d->arith;
    struct ArithReset {
        ~ArithReset() { g_arith = 0; }
    } arith_reset;
    for (int b = 0; b < d->batch; ++b) {
        const int len = lens ? lens[b] : d->t;
        conv_transpose1d_raw(x + (size_t)b * d->cin * d->t_stride, d->cin, len, d->t_stride, w, bias, d->cout, d->k, d->stride, d->crop, d->pre_slope,
                             y + (size_t)b * d->cout * d->t_out_stride, d->t_out_stride);
    }
    return 0;
}

VO_API int vo_rel_attention(int32_t batch, int32_t heads, int32_t head_dim, int32_t t, int32_t t_stride, int32_t window, const float* q, const float* k,
                            const float* v, const float* rel_k, const float* rel_v, const int32_t* lens, float* out) {
    const size_t per = (size_t)heads * head_dim * t_stride;
    for (int b = 0; b < batch; ++b)
        rel_attention(q + b * per, k + b * per, v + b * per, heads, head_dim, t, t_stride, lens ? lens[b] : t, window, rel_k, rel_v, out + b * per);
    return 0;
}

VO_API int vo_add_layer_norm(int32_t batch, int32_t channels, int32_t t, int32_t t_stride, float eps, const float* x, const float* residual,
                             const float* gamma, const float* beta, float* y) {
    Tensor g, bt;
    g.d.assign(gamma, gamma + channels);
    bt.d.assign(beta, beta + channels);
    for (int b = 0; b < batch; ++b) {
        Act a(channels, t);
        for (int c = 0; c < channels; ++c)
            for (int i = 0; i < t; ++i) {
                const size_t o = ((size_t)b * channels + c) * t_stride + i;
                a.d[(size_t)c * t + i] = x[o] + (residual ? residual[o] : 0.f);
            }
        layer_norm_channels(a, g, bt, eps);
        for (int c = 0; c < channels; ++c)
            for (int i = 0; i < t; ++i) y[((size_t)b * channels + c) * t_stride + i] = a.d[(size_t)c * t + i];
    }
    return 0;
}

// ---- helper ops (pinned by ref: test/test_ggml_utils.cpp:458-606) -------------------------------
static inline size_t idx3(const int64_t ne[3], int64_t i0, int64_t i1, int64_t i2) { return (size_t)((i2 * ne[1] + i1) * ne[0] + i0); }

/* ref: ggml-util.h:16-42. pads = {ne2_before, ne2_after, ne1_before, ne1_after, ne0_before, ne0_after} */
VO_API void vo_pad_3d(const float* src, const int64_t ne[3], const int32_t pads[6], float* dst, int64_t o[3]) {
    for (int i = 0; i < 3; ++i) {
        int ri = (3 - i - 1) * 2;
        o[i] = ne[i] + pads[ri] + pads[ri + 1];
    }
    std::fill(dst, dst + o[0] * o[1] * o[2], 0.f);
    for (int64_t k = 0; k < ne[2]; ++k)
        for (int64_t j = 0; j < ne[1]; ++j)
            for (int64_t i = 0; i < ne[0]; ++i) dst[idx3(o, i + pads[4], j + pads[2], k + pads[0])] = src[idx3(ne, i, j, k)];
}
/* ref: ggml-util.h:74-114. se = {start0,end0,start1,end1,start2,end2}; end<0 means ne+end+1 */
VO_API void vo_slice_3d(const float* src, const int64_t ne[3], const int32_t se[6], float* dst, int64_t o[3]) {
    int64_t s[3], e[3];
    for (int i = 0; i < 3; ++i) {
        s[i] = se[2 * i];
        e[i] = se[2 * i + 1];
        if (e[i] < 0) e[i] = ne[i] + (e[i] + 1);
        o[i] = e[i] - s[i];
    }
    for (int64_t k = 0; k < o[2]; ++k)
        for (int64_t j = 0; j < o[1]; ++j)
            for (int64_t i = 0; i < o[0]; ++i) dst[idx3(o, i, j, k)] = src[idx3(ne, i + s[0], j + s[1], k + s[2])];
}
/* ref: custom-ops.h:218-245 */
VO_API void vo_flip_3d(const float* src, const int64_t ne[3], int32_t along, float* dst) {
    for (int64_t k = 0; k < ne[2]; ++k)
        for (int64_t j = 0; j < ne[1]; ++j)
            for (int64_t i = 0; i < ne[0]; ++i) {
                int64_t fi = along == 0 ? ne[0] - i - 1 : i, fj = along == 1 ? ne[1] - j - 1 : j, fk = along == 2 ? ne[2] - k - 1 : k;
                dst[idx3(ne, fi, fj, fk)] = src[idx3(ne, i, j, k)];
            }
}
/* ref: ggml-util.h:163-185 */
VO_API void vo_concat_3d(const float* a, const int64_t ane[3], const float* b, const int64_t bne[3], int32_t dim, float* dst, int64_t o[3]) {
    o[0] = dim == 0 ? ane[0] + bne[0] : ane[0];
    o[1] = dim == 1 ? ane[1] + bne[1] : ane[1];
    o[2] = ane[2];
    for (int64_t k = 0; k < ane[2]; ++k)
        for (int64_t j = 0; j < ane[1]; ++j)
            for (int64_t i = 0; i < ane[0]; ++i) dst[idx3(o, i, j, k)] = a[idx3(ane, i, j, k)];
    for (int64_t k = 0; k < bne[2]; ++k)
        for (int64_t j = 0; j < bne[1]; ++j)
            for (int64_t i = 0; i < bne[0]; ++i) dst[idx3(o, i + (dim == 0 ? ane[0] : 0), j + (dim == 1 ? ane[1] : 0), k)] = b[idx3(bne, i, j, k)];
}
/* ref: custom-ops.h:329-357 */
VO_API void vo_compare(const float* a, const float* b, int64_t n, int32_t op, float* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = (op == 0 ? a[i] < b[i] : op == 1 ? a[i] >= b[i] : a[i] <= b[i]) ? 1.f : 0.f;
}
/* ref: custom-ops.h:247-273 */
VO_API void vo_per_row_cumsum(const float* src, const int64_t ne[3], float* dst) {
    for (int64_t r = 0; r < ne[1] * ne[2]; ++r) {
        float cum = 0;
        for (int64_t i = 0; i < ne[0]; ++i) {
            cum += src[r * ne[0] + i];
            dst[r * ne[0] + i] = cum;
        }
    }
}
/* ref: custom-ops.h:275-294 (running max seeded with FLT_MIN, last element) */
VO_API float vo_max(const float* src, int64_t n) {
    float cur = std::numeric_limits<float>::min();
    for (int64_t i = 0; i < n; ++i) cur = src[i] > cur ? src[i] : cur;
    return cur;
}
/* ref: custom-ops.h:885-888 */
VO_API void vo_binary_not(const float* src, int64_t n, float* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = ((int)src[i]) == 0 ? 1.f : 0.f;
}
/* ref: ggml-util.h:232-247 — view offset nb[0]*index with size_t wrap for index -1 (Q4): element index-th of
 * each row; for index = -1 the write lands one float BEFORE each row (row 0's write is out of bounds and is
 * dropped here). */
VO_API void vo_index_put_last_dim(float* t, const int64_t ne[3], int32_t index, float value) {
    for (int64_t r = 0; r < ne[1] * ne[2]; ++r) {
        int64_t pos = r * ne[0] + index;
        if (pos >= 0) t[pos] = value;
    }
}
/* ref: ggml-util.h:249-265 */
VO_API void vo_index_add_last_dim(float* t, const int64_t ne[3], int32_t index, float value) {
    for (int64_t r = 0; r < ne[1] * ne[2]; ++r) {
        int64_t pos = r * ne[0] + index;
        if (pos >= 0) t[pos] += value;
    }
}
/* ref: custom-ops.h:829-862 (consumes values sequentially where mask==1) */
VO_API void vo_masked_set(const float* t, const float* mask, const float* values, int64_t n, float* dst) {
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) dst[i] = ((int)mask[i]) == 1 ? values[k++] : t[i];
}
/* ref: custom-ops.h:739-762 as implemented: the SHAPE IS KEPT (t where the mask is 1, else 0) — what :832-840 of vits.cpp consume (Q6) */
VO_API void vo_masked_get(const float* t, const float* mask, int64_t n, float* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = ((int)mask[i]) == 1 ? t[i] : 0.f;
}
/* latents outside the spline interval met by the last vo_process_ids / vo_log_durations call of this thread (all three flows) */
VO_API int64_t vo_outside_latents(void) { return g_outside_latents; }
/* ref: test_ggml_utils.cpp:585-590 (commented out there) expects the compacted form {2,4,5}; custom-ops.h:739-762 keeps the shape
 * (zeros) instead (Q6). This is the compacted form the test vector describes. */
VO_API int64_t vo_masked_get_compact(const float* t, const float* mask, int64_t n, float* dst) {
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
        if (((int)mask[i]) == 1) dst[k++] = t[i];
    return k;
}
/* ref: custom-ops.h:764-794: dst[i] = values[index[i] + i*ne0] */
VO_API void vo_gather0(const float* t, const int64_t ne[3], const float* index, int64_t n_index, float* dst) {
    for (int64_t i = 0; i < n_index; ++i) dst[i] = t[(int64_t)index[i] + i * ne[0]];
}
/* ref: ggml-util.h:268-276 */
VO_API void vo_arange(int32_t end, float* dst) {
    for (int i = 0; i < end; ++i) dst[i] = (float)i;
}
