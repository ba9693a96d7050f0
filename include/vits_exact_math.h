/*
 * vits_exact_math.h — the ARITHMETIC of the emulated-ggml mode (vits_model_set_ggml_tables(model, 1) / vo_opts.ggml_tables = 1),
 * element by element, shared by the product (device code) and by the oracle (host code) the way vits_synth_noise.h shares the
 * definition of the synthetic inputs.
 *
 * Why it exists. In that mode GELU and the soft-max exponential go through fp16 lookup tables (ggml_gelu /
 * ggml_soft_max of the reference's ggml, /root/reference/src/vits.cpp:673,687,329,719,735 — inferred from upstream ggml, the fork is
 * absent). A table turns a last-bit difference of its argument into a 5e-4 step of its value, so two implementations that sum a
 * convolution in different orders (the MFMA chains of the throughput kernels vs the loops of the CPU oracle) — or call different
 * exp / log routines — disagree on a few durations per ten thousand ids (round 4: 5 of 8,192). The durations are the path's integer
 * output (vits.cpp:996-1001) and carry no tolerance. So in this mode stage one (text encoder + stochastic duration predictor, ~1 % of the
 * work) is computed in ONE order of operations, written down here once: every function below is a short sequential loop over IEEE-754
 * binary32 / binary64 +, -, *, /, sqrt and fma — correctly rounded on gfx950 and on x86-64 alike — and is compiled with floating-point
 * contraction OFF on both sides (csrc/Makefile, oracle/Makefile: -ffp-contract=off; the pragma below for translation units that are not).
 * No library routine is called: exp / log / softplus are the fixed polynomial evaluations vx_expf / vx_logf / vx_softplusf. The device runs
 * one thread per output element (csrc/exact_stage1.hip), the oracle one loop iteration (oracle/vits_oracle_exact.cpp): same operands,
 * same operations, same order => bit-identical log-durations, hence identical durations (GPU test: array_equal).
 * What each function restates is cited on it; the loops follow the oracle's reading of those lines (oracle/vits_oracle.cpp), which stays the
 * independent restatement and is compared with this one at tolerance (tests/test_oracle.py).
 *
 * Not used by the default mode: the throughput kernels (conv_mfma.hip, misc_kernels.hip) are unchanged.
 */
#ifndef VITS_EXACT_MATH_H
#define VITS_EXACT_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VX_HD __host__ __device__ static inline
#else
#define VX_HD static inline
#endif
#if defined(__clang__)
#define VX_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define VX_NO_CONTRACT /* g++: the translation unit is compiled with -ffp-contract=off */
#endif
#define VX_MAX_BINS 16

/* ---- bit casts, fp16 <-> fp32 in integer arithmetic (GGML_FP32_TO_FP16 / GGML_FP16_TO_FP32: round to nearest even) ---------------- */
VX_HD uint32_t vx_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
VX_HD float vx_from_bits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
VX_HD uint16_t vx_f32_to_f16(float f) {
    const uint32_t u = vx_bits(f);
    const uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                /* nan */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);               /* >= 65520 (and inf): inf */
    if (a < 0x33000001u) return (uint16_t)sign;                            /* <= 2^-25: rounds to zero (2^-25 exactly: tie to even = 0) */
    if (a < 0x38800000u) {                                                 /* fp16 subnormal: quantum 2^-24 */
        const int e = (int)(a >> 23);                                      /* biased exponent, 102 (2^-25) .. 112 (2^-15) */
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;                    /* 24-bit significand: value = m * 2^(e - 150) */
        const int sh = 126 - e;                                            /* value / 2^-24 = m >> sh, sh in 14 .. 24 */
        uint32_t q = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (q & 1u))) ++q;
        return (uint16_t)(sign | q);                                       /* (q == 0x400 is the smallest normal: the encoding carries over) */
    }
    a += 0xfffu + ((a >> 13) & 1u);                                        /* round to nearest even on bit 13 */
    return (uint16_t)(sign | ((a - 0x38000000u) >> 13));
}
VX_HD float vx_f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0x1fu) return vx_from_bits(sign | 0x7f800000u | (m << 13));
    if (e != 0u) return vx_from_bits(sign | ((e + 112u) << 23) | (m << 13));
    if (m == 0u) return vx_from_bits(sign);
    /* subnormal: m * 2^-24, exact in binary32 */
    const float v = (float)m * 5.9604644775390625e-08f;
    return vx_from_bits(vx_bits(v) | sign);
}
/* y = FP16_TO_FP32(table[FP32_TO_FP16(x)]) (ggml.c ggml_vec_gelu_f32 / ggml_compute_forward_soft_max_f32) */
VX_HD float vx_table(const uint16_t* tab, float x) { return vx_f16_to_f32(tab[vx_f32_to_f16(x)]); }

/* ---- exp, log, softplus without a library: fixed polynomials (Cephes expf / logf coefficients), fma chains ------------------------- */
VX_HD float vx_expf(float x) {
    VX_NO_CONTRACT
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float fx = x * 1.44269504088896341f;
    const int n = (int)(fx + (fx >= 0.0f ? 0.5f : -0.5f)); /* nearest integer (ties away from zero; any fixed rule does) */
    const float fn = (float)n;
    float r = fmaf(fn, -0.693359375f, x);
    r = fmaf(fn, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    return y * vx_from_bits((uint32_t)(n + 127) << 23); /* n in [-126, 127] by the clamps: a normal power of two */
}
VX_HD float vx_logf(float x) { /* x > 0, normal */
    VX_NO_CONTRACT
    uint32_t u = vx_bits(x);
    int e = (int)(u >> 23) - 126;                      /* x = m * 2^e, m in [0.5, 1) */
    float m = vx_from_bits((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m;
    }
    m = m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = p * m * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}
/* log(1 + exp(x)) (custom-ops.h:872-879), identity above 20 like the reference's op */
VX_HD float vx_softplusf(float x) {
    VX_NO_CONTRACT
    if (x > 20.0f) return x;
    const float t = vx_expf(x);
    const float u = 1.0f + t;
    if (u == 1.0f) return t;
    return vx_logf(u) * (t / (u - 1.0f)); /* log1p(t) = log(u) * t / (u - 1): compensates the rounding of 1 + t */
}

/* ---- the lookup tables, as ggml_init builds them (host only: the C library's tanhf / expf of THIS process) ----------------------------- */
static inline void vx_build_ggml_tables(uint16_t* gelu /*[65536]*/, uint16_t* exp_tab /*[65536]*/) {
    VX_NO_CONTRACT
    for (uint32_t i = 0; i < 65536u; ++i) {
        const float x = vx_f16_to_f32((uint16_t)i);
        const float inner = 1.0f + 0.044715f * x * x;
        const float arg = 0.79788456080286535587989211986876f * x * inner;
        const float g = 0.5f * x * (1.0f + tanhf(arg));
        gelu[i] = vx_f32_to_f16(g);
        exp_tab[i] = vx_f32_to_f16(expf(x));
    }
}

/* ---- convolution / Linear: one output element (vits.cpp:171-176 -> custom-ops.h:680-694; Linear :287-289,358) -------------------------- */
/* y[co][t] = bias[co] + sum over ci (outer), tap j (inner) of w[co][ci][j] * x[ci][t + j * dil - pad_l], zero outside [0, len) */
VX_HD float vx_conv_elem(const float* x, int64_t x_cs, int cin, int len, const float* w /* [cin][K] of this output channel */, float bias, int K, int dil, int pad_l,
                         int t) {
    VX_NO_CONTRACT
    float a = bias;
    for (int ci = 0; ci < cin; ++ci) {
        const float* xr = x + (int64_t)ci * x_cs;
        for (int j = 0; j < K; ++j) {
            const int tt = t + j * dil - pad_l;
            const float xv = (tt >= 0 && tt < len) ? xr[tt] : 0.0f;
            a = a + w[ci * K + j] * xv;
        }
    }
    return a;
}
/* depthwise conv of the DDS block (vits.cpp:144-169, 657-667): one channel, K taps */
VX_HD float vx_depthwise_elem(const float* xrow, int len, const float* w /*[K]*/, float bias, int K, int dil, int pad, int t) {
    VX_NO_CONTRACT
    float a = bias;
    for (int j = 0; j < K; ++j) {
        const int tt = t + j * dil - pad;
        if (tt >= 0 && tt < len) a = a + w[j] * xrow[tt];
    }
    return a;
}

/* ---- LayerNorm over channels of one column, in place (vits.cpp:115-120, 365-372, 412-418, 679-688); optional table GELU behind it ------ */
VX_HD void vx_layer_norm_column(float* col, int64_t cs, int C, const float* g, const float* b, float eps, const uint16_t* gelu_tab) {
    VX_NO_CONTRACT
    float mean = 0.0f;
    for (int c = 0; c < C; ++c) mean = mean + col[(int64_t)c * cs];
    mean = mean / (float)C;
    float var = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float dv = col[(int64_t)c * cs] - mean;
        var = var + dv * dv;
    }
    var = var / (float)C;
    const float inv = 1.0f / sqrtf(var + eps);
    for (int c = 0; c < C; ++c) {
        float v = (col[(int64_t)c * cs] - mean) * inv;
        v = v * g[c];
        v = v + b[c];
        if (gelu_tab) v = vx_table(gelu_tab, v); /* vits.cpp:673,687 ggml_gelu */
        col[(int64_t)c * cs] = v;
    }
}

/* ---- ggml_soft_max with the exp table, in place over s[0..n) (ggml.c ggml_compute_forward_soft_max_f32; vits.cpp:329,719,735) ------------- */
VX_HD void vx_softmax_table(float* s, int n, const uint16_t* exp_tab) {
    VX_NO_CONTRACT
    float mx = s[0];
    for (int i = 1; i < n; ++i) mx = s[i] > mx ? s[i] : mx;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        s[i] = vx_table(exp_tab, s[i] - mx);
        sum = sum + (double)s[i];
    }
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < n; ++i) s[i] = s[i] * inv;
}

/* ---- relative-position attention, one (head, query) (vits.cpp:296-356 with :195-235; closed form SURVEY.md App. F1) -------------------- */
/* q (scaled), k, v: this head's [hd][stride] rows; Ek / Ev [2w+1][hd]; s: scratch [len]; out: this head's [hd][stride] rows, column i */
VX_HD void vx_attention_query(const float* q, const float* k, const float* v, int64_t stride, int hd, int len, int w, const float* Ek, const float* Ev, int i, float* s,
                              const uint16_t* exp_tab, float* out) {
    VX_NO_CONTRACT
    for (int j = 0; j < len; ++j) {
        float a = 0.0f;
        for (int d = 0; d < hd; ++d) a = a + q[(int64_t)d * stride + i] * k[(int64_t)d * stride + j];
        const int r = j - i + w;
        if (r >= 0 && r <= 2 * w) {
            float bsum = 0.0f;
            for (int d = 0; d < hd; ++d) bsum = bsum + q[(int64_t)d * stride + i] * Ek[r * hd + d];
            a = a + bsum;
        }
        s[j] = a;
    }
    vx_softmax_table(s, len, exp_tab);
    for (int d = 0; d < hd; ++d) {
        float a = 0.0f;
        for (int j = 0; j < len; ++j) a = a + s[j] * v[(int64_t)d * stride + j];
        float bsum = 0.0f;
        for (int r = 0; r <= 2 * w; ++r) {
            const int j = i + r - w;
            if (j >= 0 && j < len) bsum = bsum + s[j] * Ev[r * hd + d];
        }
        out[(int64_t)d * stride + i] = a + bsum;
    }
}

/* ---- inverse rational-quadratic spline of one token's row (vits.cpp:695-802; HF modeling_vits.py:211-302) ------------------------------- */
/* uw, uh: nb unnormalised widths / heights (already / sqrt(filter_channels)); udp: nb + 1 padded unnormalised derivatives as they arrive at :704.
 * refmode: Q3 (:720). q4: reference mode, LAST token: the index -1 writes of :726,742,750 never land (ggml-util.h:235-236,252-253). */
VX_HD float vx_spline_row(float x, const float* uw, const float* uh, const float* udp, int nb, float B, int refmode, int q4, const uint16_t* exp_tab) {
    VX_NO_CONTRACT
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    float W[VX_MAX_BINS], Hh[VX_MAX_BINS], cw[VX_MAX_BINS + 1], chh[VX_MAX_BINS + 1], D[VX_MAX_BINS + 1];
    for (int i = 0; i < nb; ++i) W[i] = uw[i];
    vx_softmax_table(W, nb, exp_tab); /* :719 */
    if (refmode) {
        const float sc = min_w + (1.0f - min_w * (float)nb); /* :720 (Q3) */
        for (int i = 0; i < nb; ++i) W[i] = W[i] * sc;
    } else {
        const float sc = 1.0f - min_w * (float)nb;
        for (int i = 0; i < nb; ++i) W[i] = min_w + sc * W[i]; /* HF:225 */
    }
    float cum = 0.0f;
    cw[0] = 0.0f;
    for (int i = 0; i < nb; ++i) {
        cum = cum + W[i];
        cw[i + 1] = cum;
    }
    for (int i = 0; i <= nb; ++i) cw[i] = (B - (-B)) * cw[i] + (-B); /* :724 */
    cw[0] = -B;                                                       /* :725 */
    if (!q4) cw[nb] = B;                                              /* :726 */
    for (int i = 0; i < nb; ++i) W[i] = cw[i + 1] - cw[i];            /* :728-731 */
    for (int i = 0; i <= nb; ++i) D[i] = min_d + vx_softplusf(udp[i]); /* :733 */
    for (int i = 0; i < nb; ++i) Hh[i] = uh[i];
    vx_softmax_table(Hh, nb, exp_tab); /* :735 */
    {
        const float sc = 1.0f - min_h * (float)nb;
        for (int i = 0; i < nb; ++i) Hh[i] = min_h + sc * Hh[i]; /* :736 */
    }
    cum = 0.0f;
    chh[0] = 0.0f;
    for (int i = 0; i < nb; ++i) {
        cum = cum + Hh[i];
        chh[i + 1] = cum;
    }
    for (int i = 0; i <= nb; ++i) chh[i] = (B - (-B)) * chh[i] + (-B);
    chh[0] = -B;
    if (!q4) chh[nb] = B;
    for (int i = 0; i < nb; ++i) Hh[i] = chh[i + 1] - chh[i];
    int bin = -1; /* :748-762 */
    for (int i = 0; i <= nb; ++i) {
        float loc = chh[i];
        if (i == nb && !q4) loc = loc + 1e-6f; /* :750 */
        if (x >= loc) bin++;
    }
    bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
    const float in_cw = cw[bin], in_w = W[bin], in_ch = chh[bin], in_h = Hh[bin];
    const float delta = Hh[bin] / W[bin];
    const float d0 = D[bin], d1 = D[bin + 1];
    const float i1 = (d0 + d1) - 2.0f * delta;      /* :775 */
    const float i2 = x - in_ch;                      /* :782 */
    const float i3 = i2 * i1;                        /* :783 */
    const float a = in_h * (delta - d0) + i3;        /* :785 */
    const float b = in_h * d0 - i3;                  /* :786 */
    const float cc = (-delta) * i2;                  /* :787 */
    const float disc = b * b - (4.0f * a) * cc;      /* :789-791 */
    const float root = (2.0f * cc) / ((-b) - sqrtf(disc)); /* :792-795 */
    return root * in_w + in_cw;                      /* :797 */
}
/* the padded derivative row of one token (:826-830): pad (1, 1), both ends = log(exp(1 - min_d) - 1) — `constant` is computed ONCE on the host by
 * the caller (the C library's log / exp in double, like the reference) and handed in; reference mode: the last token's right end keeps the pad value 0 (Q4) */
VX_HD void vx_padded_derivatives(const float* ud, int64_t ud_stride, int nb, float constant, int refmode, int last_token, float* udp) {
    udp[0] = constant;
    for (int i = 0; i < nb - 1; ++i) udp[i + 1] = ud[(int64_t)i * ud_stride];
    udp[nb] = (refmode && last_token) ? 0.0f : constant;
}
/* One token of the conv flow's spline step. u: conv_proj output, rows [3 nb - 1] at stride u_cs, this token's column. Returns the value the
 * spline assigns to input xin (HF: identity outside [-B, B], HF:143-151). masked (reference mode, Q6 :837-840): the token lies outside — its
 * widths / heights / padded derivatives are zero and its input is 0. */
VX_HD float vx_spline_token(float xin, const float* u, int64_t u_cs, int nb, float B, float inv_sqrt, float constant, int refmode, int last_token, int masked,
                            const uint16_t* exp_tab) {
    VX_NO_CONTRACT
    float uw[VX_MAX_BINS], uh[VX_MAX_BINS], udp[VX_MAX_BINS + 1];
    if (masked) {
        for (int i = 0; i < nb; ++i) uw[i] = 0.0f, uh[i] = 0.0f;
        for (int i = 0; i <= nb; ++i) udp[i] = 0.0f;
    } else {
        for (int i = 0; i < nb; ++i) uw[i] = u[(int64_t)i * u_cs] * inv_sqrt;        /* :878-880 */
        for (int i = 0; i < nb; ++i) uh[i] = u[(int64_t)(nb + i) * u_cs] * inv_sqrt; /* :881-883 */
        vx_padded_derivatives(u + (int64_t)(2 * nb) * u_cs, u_cs, nb, constant, refmode, last_token, udp);
    }
    return vx_spline_row(xin, uw, uh, udp, nb, B, refmode, refmode && last_token, exp_tab);
}

/* ---- durations (vits.cpp:995-1001): ceil(exp(logw) * length_scale) ------------------------------------------------------------------------ */
VX_HD float vx_duration(float logw, float length_scale) {
    VX_NO_CONTRACT
    const float w = vx_expf(logw) * length_scale;
    if (w > 1.0e9f) return w; /* (integral already; keeps the cast below defined) */
    /* ceil without a library call: 0 < w < 2^30 */
    const float fl = (float)(int)w;
    return fl < w ? fl + 1.0f : fl;
}

#endif /* VITS_EXACT_MATH_H */
