// exact_stage1.hip — device side of the emulated-ggml mode's stage one (vits_model_set_ggml_tables(model, 1)): every kernel below is "one thread
// per output element, call the element function of include/vits_exact_math.h" — the same functions, operands and order of operations as the
// test oracle's exact-order translation unit (oracle/, never linked here), compiled with floating-point contraction off (Makefile: -ffp-contract=off), so that the log-durations —
// and with them the path's integer output, the durations (vits.cpp:996-1001) — are bit-identical on both sides. NOT a throughput path: the default
// mode's kernels (conv_mfma.hip, misc_kernels.hip) are what the benchmark runs. Layout as everywhere in stage one: [batch][channel][time], time fastest.
#include <hip/hip_runtime.h>

#include "../../include/vits_exact_math.h"
#include "kernels.h"

namespace vits {

namespace {
constexpr int NT = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + NT - 1) / NT)); }
}  // namespace

// y[b][co][t] = conv(x)[co][t] (+ relu) (* post_scale) (+ res[b][co][t]); w [cout][cin][K]
__global__ __launch_bounds__(NT) void exact_conv_kernel(const float* x, int64_t x_bs, int x_cs, const float* w, const float* bias, float* y, int64_t y_bs, int y_cs,
                                                       const float* res, int64_t r_bs, int r_cs, const int* lens, int batch, int cin, int cout, int K, int dil, int pad_l,
                                                       int tmax, int relu, int use_scale, float post_scale) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax);
    const int64_t r = idx / tmax;
    const int co = (int)(r % cout), b = (int)(r / cout);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    float v = vx_conv_elem(x + (int64_t)b * x_bs, x_cs, cin, len, w + (int64_t)co * cin * K, bias ? bias[co] : 0.0f, K, dil, pad_l, t);
    if (relu) v = v > 0.0f ? v : 0.0f;
    if (use_scale) v = v * post_scale;
    if (res) v = res[(int64_t)b * r_bs + (int64_t)co * r_cs + t] + v;
    y[(int64_t)b * y_bs + (int64_t)co * y_cs + t] = v;
}

// depthwise conv of a DDS layer: y[b][ch][t]
__global__ __launch_bounds__(NT) void exact_depthwise_kernel(const float* x, int64_t x_bs, int x_cs, const float* w, const float* bias, float* y, int64_t y_bs, int y_cs,
                                                            const int* lens, int batch, int channels, int K, int dil, int pad, int tmax) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax);
    const int64_t r = idx / tmax;
    const int ch = (int)(r % channels), b = (int)(r / channels);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    y[(int64_t)b * y_bs + (int64_t)ch * y_cs + t] = vx_depthwise_elem(x + (int64_t)b * x_bs + (int64_t)ch * x_cs, len, w + (int64_t)ch * K, bias[ch], K, dil, pad, t);
}

// x[b][c][t] += g[b][c][t]
__global__ __launch_bounds__(NT) void exact_add_kernel(float* x, int64_t x_bs, int x_cs, const float* g, int64_t g_bs, int g_cs, const int* lens, int batch, int channels,
                                                      int tmax) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax);
    const int64_t r = idx / tmax;
    const int ch = (int)(r % channels), b = (int)(r / channels);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    float* p = x + (int64_t)b * x_bs + (int64_t)ch * x_cs + t;
    *p = *p + g[(int64_t)b * g_bs + (int64_t)ch * g_cs + t];
}

// LayerNorm over channels, in place, one thread per (utterance, token); optional table GELU
__global__ __launch_bounds__(NT) void exact_layer_norm_kernel(float* x, int64_t x_bs, int x_cs, const float* gamma, const float* beta, const int* lens, int batch, int channels,
                                                             int tmax, float eps, const uint16_t* gelu_tab) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax), b = (int)(idx / tmax);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    vx_layer_norm_column(x + (int64_t)b * x_bs + t, x_cs, channels, gamma, beta, eps, gelu_tab);
}

// relative-position attention, one thread per (utterance, head, query); scratch [batch][heads][tmax][srow]
__global__ __launch_bounds__(NT) void exact_attention_kernel(const float* q, const float* k, const float* v, int64_t bs, int cs, const float* ek, const float* ev, float* out,
                                                            int64_t o_bs, int o_cs, float* scratch, int srow, const int* lens, int batch, int heads, int hd, int tmax,
                                                            int window, const uint16_t* exp_tab) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int i = (int)(idx % tmax);
    const int64_t r = idx / tmax;
    const int h = (int)(r % heads), b = (int)(r / heads);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (i >= len) return;
    const int64_t off = (int64_t)b * bs + (int64_t)h * hd * cs;
    // (q, k, v share strides: three row ranges of one buffer; `out` has its own)
    vx_attention_query(q + off, k + off, v + off, cs, hd, len, window, ek, ev, i, scratch + ((int64_t)(b * heads + h) * tmax + i) * srow, exp_tab,
                       out + (int64_t)b * o_bs + (int64_t)h * hd * o_cs);
}

// elementwise affine flow, reverse (vits.cpp:901-925): z[ch] = (z[ch] - translate[ch]) * e[ch], e = exp(+-log_scale) from the host
__global__ __launch_bounds__(NT) void exact_affine_kernel(float* z, int64_t z_bs, int z_cs, int c_first, float t0, float t1, float e0, float e1, const int* lens, int batch,
                                                         int tmax) {
    VX_NO_CONTRACT
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax);
    const int64_t r = idx / tmax;
    const int ch = (int)(r % 2), b = (int)(r / 2);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    // logical channel ch lives in physical row ch ^ c_first (the flips of vits.cpp:956 are index swaps here)
    float* p = z + (int64_t)b * z_bs + (int64_t)(ch ^ c_first) * z_cs + t;
    *p = (*p - (ch ? t1 : t0)) * (ch ? e1 : e0);
}

// spline step of a conv flow, pass 1 (one thread per token): inside mask and the spline value of every token (vits.cpp:804-852)
__global__ __launch_bounds__(NT) void exact_spline_tokens_kernel(const float* z, int64_t z_bs, int z_cs, int row, const float* u, int64_t u_bs, int u_cs, float* res,
                                                                float* inside, int stride, const int* lens, int batch, int tmax, int nb, float B, float inv_sqrt,
                                                                float constant, int refmode, const uint16_t* exp_tab) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(idx % tmax), b = (int)(idx / tmax);
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const float x = z[(int64_t)b * z_bs + (int64_t)row * z_cs + t];
    const int in = (x >= -B && x <= B) ? 1 : 0;
    inside[(int64_t)b * stride + t] = in ? 1.0f : 0.0f;
    float r;
    if (!refmode && !in) r = x;  // HF: identity outside the interval (modeling_vits.py:143-151)
    else r = vx_spline_token(in ? x : 0.0f, u + (int64_t)b * u_bs + t, u_cs, nb, B, inv_sqrt, constant, refmode, t == len - 1, !in, exp_tab);
    res[(int64_t)b * stride + t] = r;
}
// pass 2 (one thread per utterance): HF mode copies; reference mode runs the two sequential masked_set walks of :832 and :849 (Q6)
__global__ __launch_bounds__(64) void exact_spline_scatter_kernel(float* z, int64_t z_bs, int z_cs, int row, const float* res, const float* inside, float* tmp, int stride,
                                                                 const int* lens, int batch, int tmax, int refmode) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const int len = lens ? lens[b] : tmax;
    float* x1 = z + (int64_t)b * z_bs + (int64_t)row * z_cs;
    const float* rs = res + (int64_t)b * stride;
    const float* in = inside + (int64_t)b * stride;
    if (!refmode) {
        for (int t = 0; t < len; ++t) x1[t] = rs[t];
        return;
    }
    float* outv = tmp + (int64_t)b * stride;
    for (int t = 0; t < len; ++t) outv[t] = 0.0f;
    int index = 0;
    for (int t = 0; t < len; ++t)
        if (in[t] != 1.0f) {
            const int s = index++;
            outv[t] = in[s] == 1.0f ? x1[s] : 0.0f;
        }
    index = 0;
    for (int t = 0; t < len; ++t)
        if (in[t] == 1.0f) outv[t] = rs[index++];
    for (int t = 0; t < len; ++t) x1[t] = outv[t];
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
hipError_t launch_exact_conv(TensorRef x, const float* w, const float* bias, TensorRef y, TensorRef res, const int* lens, int batch, int cin, int cout, int K, int dil, int pad_l,
                             int tmax, bool relu, const float* post_scale, hipStream_t s) {
    VITS_KLAUNCH(exact_conv_kernel, grid_for((int64_t)batch * cout * tmax), dim3(NT), 0, s, x.p, x.bs, x.cs, w, bias, y.p, y.bs, y.cs, res.p, res.bs, res.cs, lens, batch, cin, cout,
                 K, dil, pad_l, tmax, relu ? 1 : 0, post_scale ? 1 : 0, post_scale ? *post_scale : 1.0f);
    return hipGetLastError();
}
hipError_t launch_exact_depthwise(TensorRef x, const float* w, const float* bias, TensorRef y, const int* lens, int batch, int channels, int K, int dil, int pad, int tmax,
                                  hipStream_t s) {
    VITS_KLAUNCH(exact_depthwise_kernel, grid_for((int64_t)batch * channels * tmax), dim3(NT), 0, s, x.p, x.bs, x.cs, w, bias, y.p, y.bs, y.cs, lens, batch, channels, K, dil, pad,
                 tmax);
    return hipGetLastError();
}
hipError_t launch_exact_add(TensorRef x, TensorRef g, const int* lens, int batch, int channels, int tmax, hipStream_t s) {
    VITS_KLAUNCH(exact_add_kernel, grid_for((int64_t)batch * channels * tmax), dim3(NT), 0, s, x.p, x.bs, x.cs, g.p, g.bs, g.cs, lens, batch, channels, tmax);
    return hipGetLastError();
}
hipError_t launch_exact_layer_norm(TensorRef x, const float* gamma, const float* beta, const int* lens, int batch, int channels, int tmax, float eps, const uint16_t* gelu_tab,
                                   hipStream_t s) {
    VITS_KLAUNCH(exact_layer_norm_kernel, grid_for((int64_t)batch * tmax), dim3(NT), 0, s, x.p, x.bs, x.cs, gamma, beta, lens, batch, channels, tmax, eps, gelu_tab);
    return hipGetLastError();
}
hipError_t launch_exact_attention(TensorRef q, TensorRef k, TensorRef v, const float* ek, const float* ev, TensorRef out, float* scratch, int srow, const int* lens, int batch,
                                  int heads, int hd, int tmax, int window, const uint16_t* exp_tab, hipStream_t s) {
    if (q.bs != k.bs || q.bs != v.bs || q.cs != k.cs || q.cs != v.cs || out.cs != q.cs) return hipErrorInvalidValue;  // (one row stride for all four, as in the oracle)
    // q, k, v are row ranges of one buffer: pass each base, the kernel adds the same offsets
    VITS_KLAUNCH(exact_attention_kernel, grid_for((int64_t)batch * heads * tmax), dim3(NT), 0, s, q.p, k.p, v.p, q.bs, q.cs, ek, ev, out.p, out.bs, out.cs, scratch, srow, lens, batch,
                 heads, hd, tmax, window, exp_tab);
    return hipGetLastError();
}
hipError_t launch_exact_affine(TensorRef z, int c_first, float t0, float t1, float e0, float e1, const int* lens, int batch, int tmax, hipStream_t s) {
    VITS_KLAUNCH(exact_affine_kernel, grid_for((int64_t)batch * 2 * tmax), dim3(NT), 0, s, z.p, z.bs, z.cs, c_first, t0, t1, e0, e1, lens, batch, tmax);
    return hipGetLastError();
}
hipError_t launch_exact_spline(TensorRef z, int row, TensorRef u, float* res, float* inside, float* tmp, int stride, const int* lens, int batch, int tmax, int nb, float B,
                               float inv_sqrt, float constant, bool refmode, const uint16_t* exp_tab, hipStream_t s) {
    if (nb > VX_MAX_BINS) return hipErrorInvalidValue;
    VITS_KLAUNCH(exact_spline_tokens_kernel, grid_for((int64_t)batch * tmax), dim3(NT), 0, s, z.p, z.bs, z.cs, row, u.p, u.bs, u.cs, res, inside, stride, lens, batch, tmax, nb, B,
                 inv_sqrt, constant, refmode ? 1 : 0, exp_tab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    VITS_KLAUNCH(exact_spline_scatter_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, z.p, z.bs, z.cs, row, res, inside, tmp, stride, lens, batch, tmax, refmode ? 1 : 0);
    return hipGetLastError();
}

}  // namespace vits
