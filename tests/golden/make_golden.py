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


@torch.no_grad()
def hf_taps(model, ids, noise_dur, noise_prior_fn):
    """Restates VitsModel.forward (modeling_vits.py:1298-1394) step by step to expose the stage outputs, with the two
    torch.randn draws replaced by the supplied arrays."""
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
        h = dec.upsampler[i](h)
        res = dec.resblocks[i * dec.num_kernels](h)
        for j in range(1, dec.num_kernels):
            res = res + dec.resblocks[i * dec.num_kernels + j](h)
        h = res / dec.num_kernels
    h = torch.nn.functional.leaky_relu(h)
    pre = dec.conv_post(h)
    wave = torch.tanh(pre)
    # cross-check against the unmodified forward of the decoder
    assert torch.allclose(wave, dec(spec), atol=1e-6)
    f = lambda t: t[0].numpy().astype(np.float32)
    return dict(
        ids=ids.astype(np.int32), noise_dur=noise_dur.astype(np.float32), noise_prior=noise_prior.astype(np.float32),
        enc_out=f(hidden), prior_mean=f(prior_means.transpose(1, 2)), prior_logvar=f(prior_logvar.transpose(1, 2)),
        log_duration=f(log_duration), durations=f(duration), z_p=f(z_p), z_flow=f(spec), pre_tanh=f(pre), waveform=f(wave),
    )


def make_ids(T, vocab, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros(T, np.int32)
    ids[1::2] = rng.integers(1, vocab, size=len(ids[1::2]))
    return ids


def taps_for(parsed, T, seed):
    model = hf_model_from_file(parsed)
    rng = np.random.default_rng(seed)
    ids = make_ids(T, model.config.vocab_size, seed)
    nd = rng.standard_normal((2, T)).astype(np.float32)
    F = model.config.flow_size
    return hf_taps(model, ids, nd, lambda L: rng.standard_normal((F, L)).astype(np.float32))


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
    print("tiny_hf_export.ggml", len(data), "bytes,", len(parsed["tensors"]), "tensors")
    # (B) tiny synthetic
    data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    np.savez(os.path.join(HERE, "tiny_synth_taps.npz"), **taps_for(parse_model_file(data), 20, 12))
    # (C) full synthetic (MMS-TTS architecture)
    data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
    taps = taps_for(parse_model_file(data), 16, 13)
    np.savez(os.path.join(HERE, "full_synth_taps.npz"), **taps)
    for k, v in taps.items():
        print(k, v.shape, float(np.sqrt((v.astype(np.float64) ** 2).mean())))


if __name__ == "__main__":
    main()
