#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/. Runs ONLY in the build container (needs transformers and,
for the exporter-written file, /root/reference/scripts/export_vits.py). Nothing here travels to the GPU box
except the data files it writes.

Fixtures (data only):
  tiny_hf_export.ggml      a tiny-architecture transformers.VitsModel written by THE REFERENCE'S OWN exporter
                           (serialize_model_to_binary + remove_weight_norm_and_convert_to_fp16,
                           /root/reference/scripts/export_vits.py:5-93). Pins every reader against the
                           reference's writer.
  tiny_hf_export_taps.npz  ids, injected noise and the stage outputs of transformers.VitsModel for that file.
  tiny_synth_taps.npz      same for the TINY synthetic model of vits_synth_model_bytes(seed, VITS_SYNTH_TINY)
  full_synth_taps.npz      same for the FULL (MMS-TTS architecture) synthetic model, T=16 ids
The stage outputs are what transformers.VitsModel — the model the reference ports (src/vits.cpp:113) and was
checked against (scripts/verify_layers.py:25) — computes; the oracle's VO_MODE_HF must reproduce them.

  *_refmode_taps.npz       the same three models and inputs through a PATCHED transformers.VitsModel that restates, in
                           torch and straight from the cited reference lines (never through the oracle), the five places
                           where /root/reference/src/vits.cpp computes something else than the model it ports (SURVEY.md
                           App. B Q1-Q5; see `reference_mode_patches`). They pin VO_MODE_REFERENCE / VITS_MODE_REFERENCE —
                           the default mode of vits_model_process and of bench.py — independently of the oracle's own reading.

  *_arith_{f16,bf16}_taps.npz  third-party pin of the 16-BIT-OPERAND arithmetic (Q7: custom-ops.h:684-690 rounds the im2col of a conv to fp16;
                           scripts/export_vits.py:79-88 stores conv weights as fp16): the same transformers model (reference-mode patches) with
                           forward-pre hooks that round the INPUT of every Conv1d / ConvTranspose1d of the flow and the vocoder to the
                           16-bit type (and the weights, for bf16), products and sums in fp32 — torch's conv, not the oracle's.
                           Pins vo_opts.arith (VO_ARITH_F16 / BF16, default scope: stage one exact) and VITS_ARITH_*.

usage: python tests/golden/make_golden.py            (from the repo root, after building csrc/libvits_hip.so)
"""
import ast
import importlib.util
import io
import os
import struct
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_package():
    spec = importlib.util.spec_from_file_location("vits_cpp_amd", os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


from modelfile_py import parse_model_file  # noqa: E402  (tests/modelfile_py.py)


def config_from_strings(cfg):
    from transformers import VitsConfig
    kw = {}
    for k, v in cfg.items():
        if k in ("model_type", "transformers_version"):
            continue
        try:
            kw[k] = ast.literal_eval(v)
        except Exception:
            kw[k] = v
    return VitsConfig(**kw)


def remove_weight_norm(model):
    import torch.nn.utils.parametrize as parametrize
    for mod in model.modules():
        if parametrize.is_parametrized(mod, "weight"):
            parametrize.remove_parametrizations(mod, "weight", leave_parametrized=True)
    return model


def hf_model_from_file(parsed):
    from transformers import VitsModel
    cfg = config_from_strings(parsed["config"])
    model = remove_weight_norm(VitsModel(cfg)).eval().float()
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, (v, _) in parsed["tensors"].items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [m for m in missing if not (m.startswith("posterior_encoder.") or ".post_" in m or ".flows.1." in m and m.startswith("duration_predictor"))]
    assert not bad, bad
    return model


# ---- reference mode: what /root/reference/src/vits.cpp computes where it deviates from transformers.VitsModel ----------------
def _put_last_wrapped(t, value, add=False):
    """index_put_last_dim / index_add_last_dim called with index = -1 (src/vits.cpp:726,742,750,830):
    `auto offset = tensor->nb[0] * index;` (src/include/ggml-util.h:235,252) multiplies a size_t by -1, so the strided
    view [1, rows] starts ONE FLOAT BEFORE the data: the write meant for the last element of row r lands on the last
    element of row r-1, row 0's write falls in front of the tensor, and the last element of the LAST row is never written.
    t: [rows = tokens, n] (the reference's ne = [n, tokens, 1])."""
    if add:
        t[:-1, -1] += value
    else:
        t[:-1, -1] = value


def _ref_rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, reverse=False, tail_bound=5.0,
                                   min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    """src/vits.cpp:695-802, statement by statement, on [tokens, bins] tensors (reverse only, like the reference :708)."""
    assert reverse
    f32 = np.float32
    upper, lower = f32(tail_bound), f32(-tail_bound)
    nb = unnormalized_widths.shape[-1]
    widths = torch.softmax(unnormalized_widths, -1)                                                   # :719
    widths = widths * float(f32(min_bin_width) + (f32(1) - f32(min_bin_width) * f32(nb)))              # :720  (Q3: the sum SCALES)
    cumwidths = torch.cumsum(widths, -1)                                                              # :721
    cumwidths = torch.nn.functional.pad(cumwidths, (1, 0))                                            # :723
    cumwidths = cumwidths * float(upper - lower) + float(lower)                                       # :724
    cumwidths[:, 0] = float(lower)                                                                    # :725
    _put_last_wrapped(cumwidths, float(upper))                                                        # :726  (Q4)
    widths = cumwidths[:, 1:] - cumwidths[:, :-1]                                                     # :728-731
    derivatives = torch.nn.functional.softplus(unnormalized_derivatives) + min_derivative             # :733
    heights = torch.softmax(unnormalized_heights, -1)                                                 # :735
    heights = heights * float(f32(1) - f32(min_bin_height) * f32(nb)) + min_bin_height                # :736
    cumheights = torch.cumsum(heights, -1)                                                            # :737
    cumheights = torch.nn.functional.pad(cumheights, (1, 0))                                          # :739
    cumheights = cumheights * float(upper - lower) + float(lower)                                     # :740
    cumheights[:, 0] = float(lower)                                                                   # :741
    _put_last_wrapped(cumheights, float(upper))                                                       # :742  (Q4)
    heights = cumheights[:, 1:] - cumheights[:, :-1]                                                  # :743-746
    bin_locations = cumheights.clone()                                                                # :748
    _put_last_wrapped(bin_locations, 1e-6, add=True)                                                  # :750  (Q4)
    bin_idx = (inputs[:, None] >= bin_locations).sum(-1) - 1                                          # :752-762
    g = lambda t: t.gather(-1, bin_idx[:, None])[:, 0]
    input_cumwidths, input_bin_widths, input_cumheights = g(cumwidths), g(widths), g(cumheights)      # :764-766
    delta = heights / widths                                                                          # :768
    input_delta = g(delta)
    input_derivatives, input_derivatives_plus_one = g(derivatives), g(derivatives[:, 1:])             # :771-772
    input_heights = g(heights)
    intermediate1 = input_derivatives + input_derivatives_plus_one - 2 * input_delta                  # :775
    intermediate2 = inputs - input_cumheights                                                         # :782
    intermediate3 = intermediate2 * intermediate1
    a = input_heights * (input_delta - input_derivatives) + intermediate3                             # :785
    b = input_heights * input_derivatives - intermediate3
    c = -input_delta * intermediate2
    discriminant = b.pow(2) - 4 * a * c                                                               # :789-791
    root = (2 * c) / (-b - torch.sqrt(discriminant))                                                  # :792-795
    return root * input_bin_widths + input_cumwidths                                                  # :797


def _masked_get(t, mask):
    """tensor_masked_get as IMPLEMENTED (src/include/custom-ops.h:739-752): `((int)src1) == 1 ? src0 : 0` element by element — the shape
    is kept, zeros where the mask is 0. (A [tokens] mask against a [tokens, n] tensor is repeated along n first: broadcast_if_possible,
    :729-736.) The reference's own known-answer test expects the compacted form and is commented out (test/test_ggml_utils.cpp:584-590)."""
    if t.dim() == 2 and mask.dim() == 1:
        mask = mask[:, None].expand_as(t)
    return torch.where(mask == 1, t, torch.zeros_like(t))


def _masked_set(t, mask, values):
    """tensor_masked_set (src/include/custom-ops.h:829-862): walks the elements in order (custom_op2, :118-141: ne[0] outermost — token
    order for the [tokens, 1, 1] tensors it is called with) and takes `values[index++]` wherever the mask is 1: the values are consumed
    SEQUENTIALLY from the front of `values`, whatever position they came from."""
    out = t.clone()
    index = 0
    for i in range(t.numel()):
        if int(mask[i]) == 1:
            out[i] = values[index]
            index += 1
    return out


def _ref_unconstrained_rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, reverse=False,
                                                 tail_bound=5.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    """src/vits.cpp:804-852, statement by statement — including the masked get / set pair of :832-849 (Q6): masked_get keeps the shape,
    masked_set consumes compacted values, so ONE latent outside [-tail_bound, tail_bound] hands every later token the spline output of
    its predecessor-by-count, and the outside tokens receive unrelated latents. With every latent inside, both masked_set calls are the
    identity permutation (the non-q6 fixtures). inputs [1, 1, T]; the others [1, 1, T, bins(-1)]."""
    assert reverse and inputs.shape[0] == 1 and inputs.shape[1] == 1
    x = inputs[0, 0]
    more_than_min = (x >= -tail_bound).float()                                                        # :819
    less_than_max = (x <= tail_bound).float()                                                         # :820
    inside = less_than_max * more_than_min                                                            # :822
    outside = 1.0 - inside                                                                            # :823 (tensor_binary_not)
    outputs = torch.zeros_like(x)                                                                     # :825
    constant = float(np.log(np.exp(1 - min_derivative) - 1))                                         # :826
    ud = torch.nn.functional.pad(unnormalized_derivatives[0, 0], (1, 1))                              # :828
    ud[:, 0] = constant                                                                               # :829
    _put_last_wrapped(ud, constant)                                                                   # :830  (Q4)
    outputs = _masked_set(outputs, outside, _masked_get(x, inside))                                   # :832  (Q6: values of the INSIDE mask)
    reshaped_inputs = _masked_get(x, inside)                                                          # :834-835
    result = _ref_rational_quadratic_spline(reshaped_inputs, _masked_get(unnormalized_widths[0, 0], inside),                    # :837-847
                                            _masked_get(unnormalized_heights[0, 0], inside), _masked_get(ud, inside), reverse, tail_bound,
                                            min_bin_width, min_bin_height, min_derivative)
    outputs = _masked_set(outputs, inside, result)                                                    # :849
    _ref_unconstrained_rational_quadratic_spline.outside_seen += int(outside.sum())
    return outputs[None, None], torch.zeros_like(inputs)


_ref_unconstrained_rational_quadratic_spline.outside_seen = 0


def _ref_elementwise_affine_forward(self, inputs, padding_mask, global_conditioning=None, reverse=False):
    """src/vits.cpp:901-925: (inputs - translate) * exp(+log_scale)  (Q5; HF: exp(-log_scale), modeling_vits.py:703)."""
    assert reverse
    return (inputs - self.translate) * torch.exp(self.log_scale) * padding_mask, None


class reference_mode_patches:
    """Context manager: transformers' VITS with the reference's Q3/Q4/Q5 arithmetic (Q1/Q2 are applied in hf_taps' decoder)."""

    def __enter__(self):
        from transformers.models.vits import modeling_vits as mv
        self.mv = mv
        self.saved = (mv._unconstrained_rational_quadratic_spline, mv.VitsElementwiseAffine.forward)
        mv._unconstrained_rational_quadratic_spline = _ref_unconstrained_rational_quadratic_spline
        mv.VitsElementwiseAffine.forward = _ref_elementwise_affine_forward
        return self

    def __exit__(self, *exc):
        self.mv._unconstrained_rational_quadratic_spline, self.mv.VitsElementwiseAffine.forward = self.saved


class conv_operand_rounding:
    """Q7 in torch: every Conv1d / ConvTranspose1d under the given sub-modules sees its input rounded (nearest even) to `dtype`, and weights rounded
    to it as well (a no-op for fp16-stored weights in fp16 mode); accumulation stays fp32. `hf_taps` rounds by hand where it calls the functional
    form (the reference-mode transposed convs)."""

    def __init__(self, modules, dtype):
        self.modules, self.dtype, self.handles, self.saved = modules, dtype, [], []

    def round(self, t):
        return t.to(self.dtype).to(torch.float32)

    def __enter__(self):
        for root in self.modules:
            for m in root.modules():
                if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d)):
                    self.handles.append(m.register_forward_pre_hook(lambda mod, args: (self.round(args[0]),) + tuple(args[1:])))
                    self.saved.append((m, m.weight.data.clone()))
                    m.weight.data = self.round(m.weight.data)
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        for m, w in self.saved:
            m.weight.data = w
        return False


@torch.no_grad()
def hf_taps(model, ids, noise_dur, noise_prior_fn, refmode=False, stage_one_only=False, rounding=None):
    """Restates VitsModel.forward (modeling_vits.py:1298-1394) step by step to expose the stage outputs, with the two
    torch.randn draws replaced by the supplied arrays. refmode: call inside `reference_mode_patches()`; additionally the
    transposed convs run without padding (Q1, src/vits.cpp:187 overwrites padding with 0), the resblock mean is a multiply by
    float(1/num_kernels) (:607,635) and the final LeakyReLU uses the config slope (Q2, :638)."""
    cfg = model.config
    input_ids = torch.from_numpy(ids.astype(np.int64))[None]
    mask = torch.ones_like(input_ids).unsqueeze(-1).float()
    enc = model.text_encoder(input_ids=input_ids, padding_mask=mask, attention_mask=None, return_dict=True)
    hidden = enc.last_hidden_state.transpose(1, 2)
    mask_t = mask.transpose(1, 2)
    prior_means, prior_logvar = enc.prior_means, enc.prior_log_variances

    real_randn = torch.randn
    try:
        torch.randn = lambda *a, **k: torch.from_numpy(noise_dur.astype(np.float32))[None]
        log_duration = model.duration_predictor(hidden, mask_t, None, reverse=True, noise_scale=model.noise_scale_duration)
    finally:
        torch.randn = real_randn
    length_scale = 1.0 / model.speaking_rate
    duration = torch.ceil(torch.exp(log_duration) * mask_t * length_scale)
    if stage_one_only:
        f = lambda t: t[0].numpy().astype(np.float32)
        return dict(ids=ids.astype(np.int32), noise_dur=noise_dur.astype(np.float32), enc_out=f(hidden), log_duration=f(log_duration), durations=f(duration))
    predicted_lengths = torch.clamp_min(torch.sum(duration, [1, 2]), 1).long()
    L = int(predicted_lengths.max())
    out_mask = (torch.arange(L)[None] < predicted_lengths[:, None]).unsqueeze(1).float()
    attn_mask = torch.unsqueeze(mask_t, 2) * torch.unsqueeze(out_mask, -1)
    b, _, out_len, in_len = attn_mask.shape
    cum = torch.cumsum(duration, -1).view(b * in_len, 1)
    idx = torch.arange(out_len, dtype=duration.dtype)
    valid = (idx.unsqueeze(0) < cum).to(attn_mask.dtype).view(b, in_len, out_len)
    padded = valid - torch.nn.functional.pad(valid, [0, 0, 1, 0, 0, 0])[:, :-1]
    attn = padded.unsqueeze(1).transpose(2, 3) * attn_mask
    pm = torch.matmul(attn.squeeze(1), prior_means).transpose(1, 2)
    plv = torch.matmul(attn.squeeze(1), prior_logvar).transpose(1, 2)
    noise_prior = noise_prior_fn(L)
    z_p = pm + torch.from_numpy(noise_prior)[None] * torch.exp(plv) * model.noise_scale
    z = model.flow(z_p, out_mask, None, reverse=True)
    spec = z * out_mask
    # decoder with a pre-tanh tap (VitsHifiGan.forward, modeling_vits.py:519-551)
    dec = model.decoder
    h = dec.conv_pre(spec)
    for i in range(dec.num_upsamples):
        h = torch.nn.functional.leaky_relu(h, cfg.leaky_relu_slope)
        up = dec.upsampler[i]
        if refmode:
            hin = rounding.round(h) if rounding is not None else h  # (functional call: the module's pre-hook does not run; its weight is already rounded)
            h = torch.nn.functional.conv_transpose1d(hin, up.weight, up.bias, stride=up.stride, padding=0)  # Q1
        else:
            h = up(h)
        res = dec.resblocks[i * dec.num_kernels](h)
        for j in range(1, dec.num_kernels):
            res = res + dec.resblocks[i * dec.num_kernels + j](h)
        h = res * float(np.float32(1.0 / dec.num_kernels)) if refmode else res / dec.num_kernels
    h = torch.nn.functional.leaky_relu(h, cfg.leaky_relu_slope) if refmode else torch.nn.functional.leaky_relu(h)  # Q2
    pre = dec.conv_post(h)
    wave = torch.tanh(pre)
    if not refmode:
        # cross-check against the unmodified forward of the decoder
        assert torch.allclose(wave, dec(spec), atol=1e-6)
    f = lambda t: t[0].numpy().astype(np.float32)
    return dict(
        ids=ids.astype(np.int32), noise_dur=noise_dur.astype(np.float32), noise_prior=noise_prior.astype(np.float32),
        enc_out=f(hidden), prior_mean=f(prior_means.transpose(1, 2)), prior_logvar=f(prior_logvar.transpose(1, 2)),
        log_duration=f(log_duration), durations=f(duration), z_p=f(z_p), z_flow=f(spec), pre_tanh=f(pre), waveform=f(wave),
    )


def arith16_taps_for(parsed, T, seed, dtype, refmode=True):
    """Same inputs as taps_for(parsed, T, seed, refmode); the flow and the vocoder in 16-bit-operand arithmetic (default scope: stage one exact fp32)."""
    model = hf_model_from_file(parsed)
    rng = np.random.default_rng(seed)
    ids = make_ids(T, model.config.vocab_size, seed)
    nd = rng.standard_normal((2, T)).astype(np.float32)
    F = model.config.flow_size
    import contextlib
    with (reference_mode_patches() if refmode else contextlib.nullcontext()):
        with conv_operand_rounding([model.flow, model.decoder], dtype) as cr:
            return hf_taps(model, ids, nd, lambda L: rng.standard_normal((F, L)).astype(np.float32), refmode=refmode, rounding=cr)


def arith16_op_fixtures():
    """Operator-level pins of the 16-bit-operand arithmetic, by torch: Conv1d / ConvTranspose1d with BOTH operands rounded to fp16 / bf16 and fp32
    accumulation (Q7: the im2col of custom-ops.h:684-690 is fp16, the exported weights are fp16, ggml accumulates in fp32), input LeakyReLU before
    the rounding where the path fuses it (vits.cpp:554,613). One set of inputs per shape, one output per arithmetic (y_f16_i, y_bf16_i)."""
    rng = np.random.default_rng(2024)
    out, n = {}, 0
    F = torch.nn.functional
    rd = lambda t, dtype: t.to(dtype).to(torch.float32)
    types = (("f16", torch.float16), ("bf16", torch.bfloat16))
    conv_cases = [(24, 40, 1, 1, 40, 1.0), (32, 32, 3, 1, 48, 0.1), (32, 32, 3, 5, 48, 0.1), (16, 24, 5, 1, 40, 1.0), (32, 32, 7, 3, 64, 0.1), (32, 32, 11, 5, 96, 0.1),
                  (192, 29, 1, 1, 17, 1.0)]
    for cin, cout, k, dil, T, slope in conv_cases:
        x = rng.standard_normal((1, cin, T)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        xt = torch.from_numpy(x)
        if slope != 1.0:
            xt = F.leaky_relu(xt, slope)
        out.update({"x_%d" % n: x, "w_%d" % n: w, "b_%d" % n: b, "meta_%d" % n: np.array([0, dil, 0, round(slope * 1e6)], np.int64)})
        for name, dtype in types:
            out["y_%s_%d" % (name, n)] = F.conv1d(rd(xt, dtype), rd(torch.from_numpy(w), dtype), torch.from_numpy(b), dilation=dil, padding=(k - 1) * dil // 2).numpy()
        n += 1
    for cin, cout, k, s, T in ((32, 16, 16, 8, 12), (32, 16, 4, 2, 40)):
        for crop in (0, (k - s) // 2):
            x = rng.standard_normal((1, cin, T)).astype(np.float32)
            w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * 2)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            out.update({"x_%d" % n: x, "w_%d" % n: w, "b_%d" % n: b, "meta_%d" % n: np.array([1, s, crop, 100000], np.int64)})
            for name, dtype in types:
                out["y_%s_%d" % (name, n)] = F.conv_transpose1d(rd(F.leaky_relu(torch.from_numpy(x), 0.1), dtype), rd(torch.from_numpy(w), dtype), torch.from_numpy(b), stride=s,
                                                                 padding=crop).numpy()
            n += 1
    out["n_cases"] = np.array([n], np.int64)
    np.savez_compressed(os.path.join(HERE, "arith16_ops.npz"), **out)
    print("arith16_ops.npz:", n, "cases x 2 arithmetics,", os.path.getsize(os.path.join(HERE, "arith16_ops.npz")), "bytes")


def make_ids(T, vocab, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros(T, np.int32)
    ids[1::2] = rng.integers(1, vocab, size=len(ids[1::2]))
    return ids


def taps_for(parsed, T, seed, refmode=False):
    model = hf_model_from_file(parsed)
    rng = np.random.default_rng(seed)
    ids = make_ids(T, model.config.vocab_size, seed)
    nd = rng.standard_normal((2, T)).astype(np.float32)
    F = model.config.flow_size
    if refmode:
        with reference_mode_patches():
            return hf_taps(model, ids, nd, lambda L: rng.standard_normal((F, L)).astype(np.float32), refmode=True)
    return hf_taps(model, ids, nd, lambda L: rng.standard_normal((F, L)).astype(np.float32))


def q6_taps_for(parsed, T, seed, noise_gain):
    """Stage one (text encoder + duration predictor) in reference mode with the duration noise scaled by `noise_gain`, so that latents
    LEAVE the spline interval [-5, 5] and the reference's masked get / set misalignment (Q6, src/vits.cpp:832-849) shapes the result.
    Stage-one taps only (an outside latent can make a token hundreds of frames long: the audio would not be a small fixture)."""
    model = hf_model_from_file(parsed)
    rng = np.random.default_rng(seed)
    ids = make_ids(T, model.config.vocab_size, seed)
    nd = (rng.standard_normal((2, T)) * noise_gain).astype(np.float32)
    with reference_mode_patches():
        _ref_unconstrained_rational_quadratic_spline.outside_seen = 0
        taps = hf_taps(model, ids, nd, None, refmode=True, stage_one_only=True)
        taps["outside_latents"] = np.array([_ref_unconstrained_rational_quadratic_spline.outside_seen], np.int64)
    assert taps["outside_latents"][0] > 0, "no latent left the interval: the fixture would not pin Q6"
    return taps


# ---- the counter-based synthetic streams of include/vits_synth_noise.h in numpy (an input DATA definition: integer hashing, one
# exact int -> float conversion, one exact subtraction and ONE rounded float multiply — bit-identical to the C header) -----------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _hash3(seed, stream, index):
    with np.errstate(over="ignore"):
        h = _mix64(np.uint64(seed) ^ np.uint64(0xD1B54A32D192ED03))
        h = _mix64(h ^ (np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)))
        return _mix64(h ^ index.astype(np.uint64))


def counter_normal(seed, stream, index):
    """vits_counter_normal(seed, stream, index) for an array of indices"""
    index = np.asarray(index, np.uint64)
    a, b = _hash3(seed, stream, np.uint64(2) * index), _hash3(seed, stream, np.uint64(2) * index + np.uint64(1))
    s = np.zeros(index.shape, np.int64)
    for h in (a, b):
        for sh in (0, 16, 32, 48):
            s += ((h >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    centred = s.astype(np.float32) - np.float32(262140.0)
    return (centred * np.float32(1.8688258e-05)).astype(np.float32)


def synth_ids(ids_seed, utt, T, vocab):
    """vits_synth_id(ids_seed, utt, t, vocab) for t in [0, T)"""
    t = np.arange(T, dtype=np.uint64)
    u = (_hash3(ids_seed + utt, 3, t) >> np.uint64(33)) % np.uint64(vocab - 1)
    ids = 1 + u.astype(np.int32)
    ids[0::2] = 0
    return ids.astype(np.int32)


def bench_utterance_taps(parsed, utt=0, T=128, ids_seed=1234, noise_seed=4321, decimate=4):
    """Utterance `utt` of bench.py's batch (64 x 128 ids: ids from the counter stream with seed 1234 + utt, noise with seed 4321 + utt)
    through the reference-mode patched transformers model: pins the benchmark's own size independently of the oracle (VERDICT r3
    missing 4). The noise is the counter stream itself (inputs need not be stored: the GPU run uses VITS_NOISE_COUNTER like the bench);
    stored: durations, log-durations, the flow output and every `decimate`-th sample of the waveform (<= 300 KB)."""
    model = hf_model_from_file(parsed)
    ids = synth_ids(ids_seed, utt, T, model.config.vocab_size)
    seed = noise_seed + utt
    nd = counter_normal(seed, 1, np.arange(2 * T)).reshape(2, T)
    F = model.config.flow_size
    with reference_mode_patches():
        _ref_unconstrained_rational_quadratic_spline.outside_seen = 0
        taps = hf_taps(model, ids, nd, lambda L: counter_normal(seed, 2, np.arange(F * L)).reshape(F, L), refmode=True)
        assert _ref_unconstrained_rational_quadratic_spline.outside_seen == 0
    return dict(ids=taps["ids"], ids_seed=np.array([ids_seed], np.int64), noise_seed=np.array([noise_seed], np.int64), utt=np.array([utt], np.int64),
                log_duration=taps["log_duration"], durations=taps["durations"], z_flow=taps["z_flow"], waveform_len=np.array([taps["waveform"].size], np.int64),
                waveform_decimated=taps["waveform"][..., ::decimate].copy(), pre_tanh_decimated=taps["pre_tanh"][..., ::decimate].copy(),
                decimate=np.array([decimate], np.int64))


def reference_exported_tiny():
    """A tiny HF model written by the reference's own exporter (scripts/export_vits.py)."""
    sys.path.insert(0, "/root/reference/scripts")
    import export_vits  # import-safe: __main__ guard at export_vits.py:95
    from transformers import VitsConfig, VitsModel
    torch.manual_seed(7)
    cfg = VitsConfig(vocab_size=38, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, window_size=2, ffn_dim=32,
                     flow_size=16, spectrogram_bins=9, upsample_initial_channel=32, upsample_rates=[4, 2],
                     upsample_kernel_sizes=[8, 4], resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3], [1, 2]],
                     depth_separable_num_layers=2, prior_encoder_num_flows=2, prior_encoder_num_wavenet_layers=2,
                     posterior_encoder_num_wavenet_layers=1)
    model = VitsModel(cfg).eval()
    # default init saturates tanh: shrink the vocoder, make the affine flow non-trivial
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("decoder.") and "weight" in n:
                p.mul_(0.6)
        model.duration_predictor.flows[0].log_scale.copy_(torch.tensor([[0.2], [-0.1]]))
        model.duration_predictor.flows[0].translate.copy_(torch.tensor([[-0.5], [0.3]]))
    model = export_vits.remove_weight_norm_and_convert_to_fp16(model)

    class Tok:  # the attributes serialize_model_to_binary reads (export_vits.py:8-27)
        phonemize = False
        is_uroman = False
        add_blank = True
        normalize = True
        pad_token = "<pad>"
        unk_token = "<unk>"

        def get_vocab(self):
            v = {"<pad>": 0, " ": 1, "'": 2, "-": 3}
            for i, c in enumerate("abcdefghijklmnopqrstuvwxyz"):
                v[c] = 4 + i
            for i, c in enumerate("0123456"):
                v[c] = 30 + i
            v["<unk>"] = 37
            return v

    path = os.path.join(HERE, "tiny_hf_export.ggml")
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        export_vits.serialize_model_to_binary(model.config, model.state_dict(), Tok(), path)
    return open(path, "rb").read()


def main():
    pkg = load_package()
    # (A) exporter-written tiny HF model
    data = reference_exported_tiny()
    parsed = parse_model_file(data)
    np.savez(os.path.join(HERE, "tiny_hf_export_taps.npz"), **taps_for(parsed, 14, 11))
    np.savez(os.path.join(HERE, "tiny_hf_export_refmode_taps.npz"), **taps_for(parsed, 14, 11, refmode=True))
    print("tiny_hf_export.ggml", len(data), "bytes,", len(parsed["tensors"]), "tensors")
    # (B) tiny synthetic
    data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    np.savez(os.path.join(HERE, "tiny_synth_taps.npz"), **taps_for(parse_model_file(data), 20, 12))
    np.savez(os.path.join(HERE, "tiny_synth_refmode_taps.npz"), **taps_for(parse_model_file(data), 20, 12, refmode=True))
    # (C) full synthetic (MMS-TTS architecture)
    data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
    taps = taps_for(parse_model_file(data), 16, 13)
    np.savez(os.path.join(HERE, "full_synth_taps.npz"), **taps)
    rtaps = taps_for(parse_model_file(data), 16, 13, refmode=True)
    np.savez(os.path.join(HERE, "full_synth_refmode_taps.npz"), **rtaps)
    print("reference mode: frames", int(rtaps["durations"].sum()), "vs HF", int(taps["durations"].sum()), "; samples", rtaps["waveform"].size, "vs", taps["waveform"].size)
    for k, v in taps.items():
        print(k, v.shape, float(np.sqrt((v.astype(np.float64) ** 2).mean())))
    # (C2) the two lengths SURVEY section 7 step 1 names (T = 8 and 32 ids), full architecture, both modes
    for T, seed in ((8, 31), (32, 32)):
        np.savez_compressed(os.path.join(HERE, "full_synth_T%d_taps.npz" % T), **taps_for(parse_model_file(data), T, seed))
        np.savez_compressed(os.path.join(HERE, "full_synth_T%d_refmode_taps.npz" % T), **taps_for(parse_model_file(data), T, seed, refmode=True))
    # (C3) the 16-bit-operand arithmetic modes (Q7), reference mode, tiny + full architecture, same inputs as (B) / (C)
    tiny = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    for name, dtype in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        a = arith16_taps_for(parse_model_file(tiny), 20, 12, dtype)
        np.savez_compressed(os.path.join(HERE, "tiny_synth_arith_%s_taps.npz" % name), **a)
        b16 = arith16_taps_for(parse_model_file(data), 16, 13, dtype)
        np.savez_compressed(os.path.join(HERE, "full_synth_arith_%s_taps.npz" % name), **b16)
        d = b16["waveform"].astype(np.float64) - rtaps["waveform"]
        print("arith", name, "full: waveform vs the fp32 reference-mode fixture: max %.2e rms %.2e of RMS" % (
            np.abs(d).max() / np.sqrt((rtaps["waveform"].astype(np.float64) ** 2).mean()), np.sqrt((d ** 2).mean()) / np.sqrt((rtaps["waveform"].astype(np.float64) ** 2).mean())))
    arith16_op_fixtures()
    # (D) Q6: latents outside the spline interval (duration noise x 4: |0.8 * 4 * n| > 5 for one draw in eight), stage one only
    q = q6_taps_for(parse_model_file(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)), 24, 21, 4.0)
    np.savez(os.path.join(HERE, "tiny_synth_q6_refmode_taps.npz"), **q)
    print("tiny q6: outside latents", int(q["outside_latents"][0]), "durations", q["durations"].astype(int).ravel().tolist())
    q = q6_taps_for(parse_model_file(data), 40, 22, 4.0)
    np.savez(os.path.join(HERE, "full_synth_q6_refmode_taps.npz"), **q)
    print("full q6: outside latents", int(q["outside_latents"][0]), "durations", q["durations"].astype(int).ravel().tolist())
    # (E) the benchmark's own size: utterance 0 of bench.py's batch (128 ids) through the patched model
    b = bench_utterance_taps(parse_model_file(data))
    np.savez_compressed(os.path.join(HERE, "bench_utt0_refmode_taps.npz"), **b)
    print("bench utterance 0: frames", int(b["durations"].sum()), "samples", int(b["waveform_len"][0]), "file",
          os.path.getsize(os.path.join(HERE, "bench_utt0_refmode_taps.npz")), "bytes")


if __name__ == "__main__":
    main()
