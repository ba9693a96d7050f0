"""CPU tests that PIN THE ORACLE (no GPU): golden taps from transformers.VitsModel, the reference exporter's own
file, the reference's helper-op known-answer vectors and its libstdc++ noise stream."""
import os

import numpy as np
import pytest

from conftest import golden, rel_err
from modelfile_py import parse_model_file

FLOAT_TAPS = ["enc_out", "prior_mean", "prior_logvar", "log_duration", "z_p", "z_flow", "pre_tanh", "waveform"]


def _bytes_for(name, pkg, tiny_hf_bytes):
    if name.startswith("tiny_hf_export"):
        return tiny_hf_bytes
    return pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY if name.startswith("tiny") else pkg.SYNTH_FULL)


@pytest.mark.parametrize("fixture", ["tiny_hf_export_taps.npz", "tiny_synth_taps.npz", "full_synth_taps.npz", "full_synth_T8_taps.npz", "full_synth_T32_taps.npz"])
def test_oracle_hf_mode_reproduces_transformers_taps(pkg, oracle, tiny_hf_bytes, fixture):
    """oracle(VO_MODE_HF) == transformers.VitsModel stage by stage (<= 1e-4 of each tap's RMS; durations exact).
    tiny_hf_export.ggml was written by the REFERENCE'S OWN exporter (scripts/export_vits.py), so this also pins the reader."""
    g = golden(fixture)
    m = oracle.Model(_bytes_for(fixture, pkg, tiny_hf_bytes))
    r = m.process_ids(g["ids"], mode=oracle.MODE_HF, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=g["noise_dur"], noise_prior=g["noise_prior"])
    np.testing.assert_array_equal(r["durations"], g["durations"].ravel())
    for name in FLOAT_TAPS:
        assert rel_err(r[name], g[name]) < 1e-4, name


@pytest.mark.parametrize("fixture", ["tiny_hf_export_refmode_taps.npz", "tiny_synth_refmode_taps.npz", "full_synth_refmode_taps.npz", "full_synth_T8_refmode_taps.npz",
                                     "full_synth_T32_refmode_taps.npz"])
def test_oracle_reference_mode_reproduces_the_patched_transformers_taps(pkg, oracle, tiny_hf_bytes, fixture):
    """oracle(VO_MODE_REFERENCE) == a transformers.VitsModel PATCHED with torch restatements of the reference lines where
    vits.cpp deviates from the model it ports (Q1 vits.cpp:187, Q2 :638, Q3 :720, Q4 ggml-util.h:235,252 via vits.cpp:726,742,750,830,
    Q5 :913-918; tests/golden/make_golden.py `reference_mode_patches`) — written from the reference source, not from the oracle.
    This is what pins the DEFAULT mode of vits_model_process / bench.py independently of the oracle's own reading."""
    g = golden(fixture)
    h = golden(fixture.replace("_refmode", ""))
    m = oracle.Model(_bytes_for(fixture, pkg, tiny_hf_bytes))
    r = m.process_ids(g["ids"], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=g["noise_dur"], noise_prior=g["noise_prior"])
    np.testing.assert_array_equal(r["durations"], g["durations"].ravel())
    for name in FLOAT_TAPS:
        assert rel_err(r[name], g[name]) < 1e-4, name
    # the fixture really exercises the deviations: same inputs, different durations and the uncropped length (Q1)
    np.testing.assert_array_equal(g["ids"], h["ids"])
    assert not np.array_equal(g["durations"], h["durations"]) or not np.allclose(g["log_duration"], h["log_duration"])
    ups = 1
    for s_ in eval(m.config("upsample_rates") or "[8, 8, 2, 2]"):
        ups *= s_
    assert g["waveform"].size > ups * int(g["durations"].sum())


@pytest.mark.parametrize("fixture", ["tiny_synth_q6_refmode_taps.npz", "full_synth_q6_refmode_taps.npz"])
def test_oracle_reference_mode_reproduces_the_masked_get_set_misalignment(pkg, oracle, tiny_hf_bytes, fixture):
    """Q6 (VERDICT r3 missing 1): with latents OUTSIDE [-5, 5] the reference's tensor_masked_get keeps the shape while tensor_masked_set
    consumes compacted values (vits.cpp:832-849, custom-ops.h:739-752,829-862): one outside latent hands every later token its
    predecessor-by-count's spline output. The fixture is a transformers.VitsModel patched with a statement-by-statement torch restatement
    of those lines (make_golden.py, duration noise x 4); oracle(VO_MODE_REFERENCE) must reproduce its log-durations and durations — and
    HF's identity tails (VO_MODE_HF, and the oracle's reading before round 4) must NOT."""
    g = golden(fixture)
    m = oracle.Model(_bytes_for(fixture, pkg, tiny_hf_bytes))
    logw, dur = m.log_durations(g["ids"], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=g["noise_dur"])
    assert oracle.outside_latents() == int(g["outside_latents"][0]) > 0
    np.testing.assert_array_equal(dur, g["durations"].ravel())
    assert rel_err(logw, g["log_duration"]) < 1e-4
    logw_hf, _ = m.log_durations(g["ids"], mode=oracle.MODE_HF, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=g["noise_dur"])
    assert rel_err(logw_hf, g["log_duration"]) > 1e-2


def test_counter_noise_numpy_port_is_the_header(pkg, oracle):
    """make_golden.py's numpy port of include/vits_synth_noise.h (used to feed the patched transformers model the benchmark's counter
    noise) is bit-identical to the C header: the oracle run on the counter stream equals the oracle run on the port's values."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY))
    for utt in (0, 5):
        ids = mg.synth_ids(1234, utt, 31, 38)
        np.testing.assert_array_equal(ids, pkg.synth_ids(utt + 1, 31)[utt])
        nd = mg.counter_normal(4321 + utt, 1, np.arange(2 * 31)).reshape(2, 31)
        a = m.log_durations(ids, noise_kind=oracle.NOISE_COUNTER, noise_seed=4321 + utt)
        b = m.log_durations(ids, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=nd)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    assert abs(float(mg.counter_normal(7, 2, np.arange(200000)).std()) - 1.0) < 0.01


def test_oracle_reproduces_the_benchmark_size_utterance(pkg, oracle):
    """VERDICT r3 missing 4: utterance 0 of bench.py's batch (128 ids, ids seed 1234, counter noise seed 4321) through the patched
    transformers model (tests/golden/bench_utt0_refmode_taps.npz) — the benchmark's own size pinned independently of the oracle."""
    g = golden("bench_utt0_refmode_taps.npz")
    ids = pkg.synth_ids(1, 128, ids_seed=int(g["ids_seed"][0]))[0]
    np.testing.assert_array_equal(ids, g["ids"])
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
    r = m.process_ids(ids, mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_COUNTER, noise_seed=int(g["noise_seed"][0]) + int(g["utt"][0]),
                      taps=["log_duration", "durations", "z_flow", "pre_tanh", "waveform"])
    np.testing.assert_array_equal(r["durations"], g["durations"].ravel())
    assert oracle.outside_latents() == 0
    assert r["waveform"].size == int(g["waveform_len"][0])
    d = int(g["decimate"][0])
    assert rel_err(r["log_duration"], g["log_duration"]) < 1e-4
    assert rel_err(r["z_flow"], g["z_flow"]) < 1e-4
    assert rel_err(r["pre_tanh"][::d], g["pre_tanh_decimated"]) < 2e-4
    assert rel_err(r["waveform"][::d], g["waveform_decimated"]) < 2e-4


def test_readers_agree_with_reference_exporter_file(pkg, oracle, tiny_hf_bytes):
    """Three readers (product C++, oracle C++, numpy) on the file the reference's exporter wrote."""
    py = parse_model_file(tiny_hf_bytes)
    om = oracle.Model(tiny_hf_bytes)
    assert om.num_tensors() == len(py["tensors"]) == 314
    for name, (arr, dt) in py["tensors"].items():
        t, odt = om.tensor(name)
        assert odt == dt and t.shape == arr.shape
        np.testing.assert_array_equal(t, arr.astype(np.float32))  # fp16 -> fp32 widening is exact
    assert om.config("hidden_size") == py["config"]["hidden_size"] == "16"
    assert om.config("upsample_rates") == "[4, 2]"
    # export_vits.py:79-88 stores every Conv1d / ConvTranspose1d weight as fp16, everything else fp32
    assert py["tensors"]["decoder.upsampler.0.weight"][1] == 1 and py["tensors"]["text_encoder.embed_tokens.weight"][1] == 0
    assert py["tensors"]["decoder.upsampler.0.weight"][0].shape == (32, 16, 8)  # torch ConvTranspose1d [Cin][Cout][K]
    # product reader + writer: byte-exact round trip of the reference's format
    assert pkg.reserialize(tiny_hf_bytes) == tiny_hf_bytes


def test_synthetic_model_is_deterministic_and_well_formed(pkg, oracle):
    a = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    b = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    c = pkg.synth_model_bytes(0x5EEE, pkg.SYNTH_TINY)
    assert a == b and a != c
    assert pkg.reserialize(a) == a
    py = parse_model_file(a)
    assert py["add_blank"] == 1 and py["pad"] == "<pad>" and len(py["vocab"]) == 38
    w, dt = py["tensors"]["flow.flows.0.wavenet.in_layers.0.weight"]
    assert dt == 1 and w.shape == (32, 16, 5)


def test_reference_mode_applies_the_documented_deviations(pkg, oracle):
    """VO_MODE_REFERENCE vs VO_MODE_HF on the same inputs: encoder identical, durations differ through Q3-Q5, the vocoder
    output has the uncropped length S = 256 L + 294 (Q1, vits.cpp:187)."""
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
    ids = pkg.synth_ids(1, 10)[0]
    hf = m.process_ids(ids, mode=oracle.MODE_HF, noise_seed=3)
    rf = m.process_ids(ids, mode=oracle.MODE_REFERENCE, noise_seed=3)
    np.testing.assert_array_equal(hf["enc_out"], rf["enc_out"])
    assert not np.allclose(hf["log_duration"], rf["log_duration"])
    L_hf, L_rf = int(hf["durations"].sum()), int(rf["durations"].sum())
    assert hf["waveform"].size == 256 * L_hf
    assert rf["waveform"].size == 256 * L_rf + 294
    # pinned durations: identical latents, so the two vocoders differ only by Q1/Q2
    hf2 = m.process_ids(ids, mode=oracle.MODE_HF, noise_seed=3, fixed_duration=2)
    rf2 = m.process_ids(ids, mode=oracle.MODE_REFERENCE, noise_seed=3, fixed_duration=2)
    np.testing.assert_array_equal(hf2["z_flow"], rf2["z_flow"])
    assert rf2["waveform"].size - hf2["waveform"].size == 294


def test_reference_noise_stream_known_answer(oracle):
    """libstdc++ default_random_engine (minstd_rand0, seed 1) + normal_distribution<float>: the stream the reference draws
    from (vits.cpp:31, ggml-util.h:187-199). Known answer: SURVEY.md §8c (g++ 11.4)."""
    first = oracle.reference_noise(8, seed=1)
    expect = np.array([-0.259093195, 1.60159206, -1.49896121, 0.174767554, 0.119264036, -0.302023172, 0.458181173, 0.188984558], np.float32)
    np.testing.assert_allclose(first, expect, rtol=0, atol=1e-7)
    again = oracle.reference_noise(8)  # never reseeded: the stream continues
    assert not np.allclose(first, again)


def test_tokenizer_lowercases_matches_vocab_and_intersperses_blanks(pkg, oracle, tiny_bytes):
    m = oracle.Model(tiny_bytes)
    ids = m.tokenize("Hi, a-b!")  # ',' and '!' are not in the vocabulary and are dropped (vits_tokenizer.cpp:72-75)
    vocab = parse_model_file(tiny_bytes)["vocab"]
    want = [vocab[c] for c in "hi a-b"]
    assert list(ids[1::2]) == want and set(ids[0::2]) == {vocab["<pad>"]} and len(ids) == 2 * len(want) + 1
    np.testing.assert_array_equal(pkg.file_tokenize(tiny_bytes, "Hi, a-b!"), ids)  # product tokenizer (host only)


# ---- the reference's helper-op known-answer vectors (test/test_ggml_utils.cpp:458-606) -------------------------
import ctypes as C  # noqa: E402


def _i64(*v):
    return (C.c_int64 * len(v))(*v)


def _i32(*v):
    return (C.c_int32 * len(v))(*v)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


INPUT = np.array([1, 2, 3, 4, 5, 6], np.float32)  # ne = [3, 2, 1]


@pytest.mark.parametrize("pads,expected,shape", [
    ((0, 0, 0, 0, 0, 0), [1, 2, 3, 4, 5, 6], (3, 2, 1)),
    ((0, 0, 0, 2, 0, 0), [1, 2, 3, 4, 5, 6, 0, 0, 0, 0, 0, 0], (3, 4, 1)),
    ((0, 0, 2, 0, 0, 0), [0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6], (3, 4, 1)),
    ((0, 0, 2, 1, 0, 0), [0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 0, 0, 0], (3, 5, 1)),
    ((0, 0, 0, 0, 0, 2), [1, 2, 3, 0, 0, 4, 5, 6, 0, 0], (5, 2, 1)),
    ((0, 0, 0, 0, 3, 0), [0, 0, 0, 1, 2, 3, 0, 0, 0, 4, 5, 6], (6, 2, 1)),
    ((1, 0, 0, 0, 0, 0), [0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6], (3, 2, 2)),
    ((0, 1, 0, 0, 0, 0), [1, 2, 3, 4, 5, 6, 0, 0, 0, 0, 0, 0], (3, 2, 2)),
])
def test_kat_pad_3d(oracle, pads, expected, shape):  # test_ggml_utils.cpp:458-465
    out = np.zeros(len(expected), np.float32)
    one = _i64(0, 0, 0)
    oracle.lib().vo_pad_3d(_p(INPUT), _i64(3, 2, 1), _i32(*pads), _p(out), one)
    assert tuple(one) == shape and out.tolist() == expected


@pytest.mark.parametrize("src,ne,se,expected,shape", [
    (INPUT, (3, 2, 1), (0, -1, 0, -1, 0, -1), [1, 2, 3, 4, 5, 6], (3, 2, 1)),
    (INPUT, (3, 2, 1), (0, -1, 0, -1, 0, 1), [1, 2, 3, 4, 5, 6], (3, 2, 1)),
    (INPUT, (3, 2, 1), (0, -1, 0, 1, 0, -1), [1, 2, 3], (3, 1, 1)),
    (INPUT, (3, 2, 1), (0, -1, 1, -1, 0, -1), [4, 5, 6], (3, 1, 1)),
    (INPUT, (3, 2, 1), (0, 2, 0, -1, 0, -1), [1, 2, 4, 5], (2, 2, 1)),
    (INPUT, (3, 2, 1), (2, -1, 0, -1, 0, -1), [3, 6], (1, 2, 1)),
    (np.arange(1, 17, dtype=np.float32), (2, 4, 2), (0, -1, 0, 1, 0, -1), [1, 2, 9, 10], (2, 1, 2)),
    (np.arange(1, 17, dtype=np.float32), (2, 4, 2), (0, -1, 1, 2, 0, -1), [3, 4, 11, 12], (2, 1, 2)),
    (np.arange(1, 17, dtype=np.float32), (2, 4, 2), (0, -1, 2, 3, 0, -1), [5, 6, 13, 14], (2, 1, 2)),
    (np.arange(1, 17, dtype=np.float32), (2, 4, 2), (0, -1, 3, 4, 0, -1), [7, 8, 15, 16], (2, 1, 2)),
])
def test_kat_slice_3d(oracle, src, ne, se, expected, shape):  # test_ggml_utils.cpp:469-485 (split :489-491 is two slices)
    out = np.zeros(len(expected), np.float32)
    one = _i64(0, 0, 0)
    oracle.lib().vo_slice_3d(_p(src), _i64(*ne), _i32(*se), _p(out), one)
    assert tuple(one) == shape and out.tolist() == expected


def test_kat_flip_concat_compare_cumsum_max_not(oracle):
    L = oracle.lib()
    out = np.zeros(6, np.float32)
    L.vo_flip_3d(_p(INPUT), _i64(3, 2, 1), 0, _p(out))  # :494
    assert out.tolist() == [3, 2, 1, 6, 5, 4]
    L.vo_flip_3d(_p(INPUT), _i64(3, 2, 1), 1, _p(out))  # :495
    assert out.tolist() == [4, 5, 6, 1, 2, 3]
    a, b = np.array([1, 2, 3], np.float32), np.array([1, 7, 8], np.float32)
    one = _i64(0, 0, 0)
    L.vo_concat_3d(_p(b), _i64(3, 1, 1), _p(a), _i64(3, 1, 1), 1, _p(out), one)  # :500
    assert out.tolist() == [1, 7, 8, 1, 2, 3] and tuple(one) == (3, 2, 1)
    L.vo_concat_3d(_p(b), _i64(3, 1, 1), _p(a), _i64(3, 1, 1), 0, _p(out), one)  # :502
    assert out.tolist() == [1, 7, 8, 1, 2, 3] and tuple(one) == (6, 1, 1)
    o3 = np.zeros(3, np.float32)
    L.vo_compare(_p(a), _p(b), 3, 0, _p(o3))  # a < b  :514
    assert o3.tolist() == [0, 1, 1]
    L.vo_compare(_p(a), _p(b), 3, 1, _p(o3))  # a >= b :516
    assert o3.tolist() == [1, 0, 0]
    L.vo_per_row_cumsum(_p(a), _i64(3, 1, 1), _p(o3))  # :532-533
    assert o3.tolist() == [1, 3, 6]
    c = np.array([6, 2, 3, 8, 4, 2], np.float32)
    assert L.vo_max(_p(c), 6) == 8.0  # :535-536
    n = np.array([0, 1, 0, 1, 0, 0], np.float32)
    L.vo_binary_not(_p(n), 6, _p(out))  # :538
    assert out.tolist() == [1, 0, 1, 0, 1, 1]


def test_kat_index_put_add_masked_set_arange(oracle):
    L = oracle.lib()
    for index, value, expected in [(0, 10, [10, 2, 3, 10, 5, 6]), (2, 10, [1, 2, 10, 4, 5, 10]), (1, 5, [1, 5, 3, 4, 5, 6])]:  # :540-565
        t = INPUT.copy()
        L.vo_index_put_last_dim(_p(t), _i64(3, 2, 1), index, C.c_float(value))
        assert t.tolist() == expected
    t = INPUT.copy()
    L.vo_index_add_last_dim(_p(t), _i64(3, 2, 1), 1, C.c_float(9))  # :568-575
    assert t.tolist() == [1, 11, 3, 4, 14, 6]
    # Q4: index -1 wraps to "one float before each row" (ggml-util.h:235-236): row r's write lands on row r-1's last
    # element, the last row's own last element is never written (and row 0's write is out of bounds, dropped here)
    t = INPUT.copy()
    L.vo_index_put_last_dim(_p(t), _i64(3, 2, 1), -1, C.c_float(10))
    assert t.tolist() == [1, 2, 10, 4, 5, 6]
    mask = np.array([0, 1, 0, 1, 1, 0], np.float32)
    vals = np.full(6, 10, np.float32)
    out = np.zeros(6, np.float32)
    L.vo_masked_set(_p(INPUT), _p(mask), _p(vals), 6, _p(out))  # :577-583
    assert out.tolist() == [1, 10, 3, 10, 10, 6]
    o3 = np.zeros(6, np.float32)
    assert L.vo_masked_get_compact(_p(INPUT), _p(mask), 6, _p(o3)) == 3 and o3[:3].tolist() == [2, 4, 5]  # :585-590 (COMMENTED OUT in the reference:
    # its masked_get keeps the shape — custom-ops.h:746-749 `((int)src1) == 1 ? src0 : 0` — which is what vits.cpp:832-840 consume, Q6)
    o4 = np.full(6, -1, np.float32)
    L.vo_masked_get(_p(INPUT), _p(mask), 6, _p(o4))
    assert o4.tolist() == [0, 2, 0, 4, 5, 0]
    ar = np.zeros(6, np.float32)
    L.vo_arange(6, _p(ar))  # :600-605
    assert ar.tolist() == [0, 1, 2, 3, 4, 5]


def test_bf16_storage_extension_round_trips(pkg, oracle):
    """Type tag 2 (bf16) is this repo's extension for BASELINE.json config 5 ("bf16 weights"): same tensors, bf16-rounded."""
    a = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)
    b = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY | pkg.SYNTH_BF16)
    assert len(a) == len(b) and a != b and pkg.reserialize(b) == b
    ma, mb = oracle.Model(a), oracle.Model(b)
    wa, ta = ma.tensor("decoder.conv_pre.weight")
    wb, tb = mb.tensor("decoder.conv_pre.weight")
    assert (ta, tb) == (1, 2)
    assert np.abs(wa - wb).max() <= np.abs(wa).max() * 2.0 ** -8  # bf16 has 8 mantissa bits
    ea, _ = ma.tensor("text_encoder.embed_tokens.weight")
    eb, _ = mb.tensor("text_encoder.embed_tokens.weight")
    np.testing.assert_array_equal(ea, eb)  # fp32 tensors untouched


def test_arith_scope_keeps_stage_one_exact(pkg, oracle, tiny_bytes):
    """vo_opts.arith_scope: under FLOW_VOCODER a 16-bit arithmetic leaves the text encoder and the duration predictor in exact fp32
    (every stage-one tap and the integer durations equal the fp32 run bit for bit; the flow output and the waveform move), under
    ALL_CONVS the log-durations move too (custom-ops.h:684-690 rounds the im2col of every conv, the duration predictor's included)."""
    om = oracle.Model(tiny_bytes)
    ids = np.zeros(21, np.int32)
    ids[1::2] = np.arange(1, 11)
    base = om.process_ids(ids, noise_seed=3)
    for arith in (oracle.ARITH_F16, oracle.ARITH_BF16):
        fv = om.process_ids(ids, noise_seed=3, arith=arith, arith_scope=oracle.SCOPE_FLOW_VOCODER)
        for tap in ("enc_out", "prior_mean", "prior_logvar", "log_duration", "durations", "z_p"):
            assert np.array_equal(fv[tap], base[tap]), tap
        assert fv["waveform"].size == base["waveform"].size and not np.array_equal(fv["z_flow"], base["z_flow"])
        assert not np.array_equal(fv["waveform"], base["waveform"])
        ac = om.process_ids(ids, noise_seed=3, arith=arith, arith_scope=oracle.SCOPE_ALL_CONVS, fixed_duration=2)
        b2 = om.process_ids(ids, noise_seed=3, fixed_duration=2)
        assert not np.array_equal(ac["log_duration"], b2["log_duration"]) and not np.array_equal(ac["enc_out"], b2["enc_out"])


def test_oracle_is_clean_under_asan_and_ubsan():
    """The reference's Debug configuration is an AddressSanitizer build (/root/reference/CMakeLists.txt:9-13). The GPU pool has no
    device-side sanitizer, so the CPU restatement gets it: `make -C oracle asan` builds the oracle + a driver with
    -fsanitize=address,undefined, and the driver runs the tiny exporter-written model through every mode / arithmetic / short
    input, the reference noise stream, a truncated file and the index -1 wrap of Q4 (an out-of-bounds write in the reference,
    ggml-util.h:235 — the restatement must model its effect without one)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(root, "oracle", "_asan", "asan_driver"), os.path.join(root, "tests", "golden", "tiny_hf_export.ggml")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "asan driver ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_emulated_ggml_tables_option(pkg, oracle):
    """vo_opts.ggml_tables (Q8; INFERRED from upstream ggml, the reference's fork is absent): tanh-GELU through an fp16 table and the
    soft-max exponential through an fp16 table with a double sum. Off by default (every fixture is in the default arithmetic); on, the
    log-durations move by ~1e-3..1e-2 and stay finite; the table functions themselves against numpy's float16 rounding."""
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
    ids = pkg.synth_ids(2, 64)
    for u in range(2):
        a = m.log_durations(ids[u], noise_seed=4321 + u)
        b = m.log_durations(ids[u], noise_seed=4321 + u, ggml_tables=True)
        c = m.log_durations(ids[u], noise_seed=4321 + u)
        np.testing.assert_array_equal(a[0], c[0])  # the switch does not stick
        d = np.abs(a[0] - b[0]).max()
        assert 1e-5 < d < 5e-2 and np.isfinite(b[0]).all()
        assert (a[1] != b[1]).mean() < 0.05


@pytest.mark.parametrize("arch", ["tiny", "full"])
@pytest.mark.parametrize("name,tol", [("f16", 5e-3), ("bf16", 8e-2)])
def test_oracle_16bit_arithmetic_reproduces_the_torch_operand_rounding_fixtures(pkg, oracle, arch, name, tol):
    """Third-party pin of vo_opts.arith ("the 16-bit-mode oracle has no third-party pin at all"): the fixtures are
    transformers.VitsModel (reference-mode patches) with forward-pre hooks that round the input — and, for bf16, the weights — of every
    Conv1d / ConvTranspose1d of the flow and the vocoder to the 16-bit type, fp32 products and sums by torch's own conv
    (tests/golden/make_golden.py `conv_operand_rounding`; Q7: /root/reference/src/include/custom-ops.h:684-690 + scripts/export_vits.py:79-88).
    Stage one is exact fp32 under the default scope: its taps and the integer durations meet the fp32 bounds.
    Downstream, 16-bit-operand arithmetic is CHAOTIC at the size of its own rounding step: two correct implementations differ by fp32 summation
    order, an activation within 1e-7 of a rounding boundary then rounds the other way (1e-3 relative for fp16, 8e-3 for bf16) and the gated
    WaveNet layers carry that on. Measured inside torch alone (fp32 against fp64 accumulation of the SAME rounded operands, full architecture):
    z_flow differs by 5.2e-4 (fp16) / 3.3e-3 (bf16) of RMS — exactly what the oracle shows against the fixture. So the whole-model bound is the
    one of the GPU-vs-oracle tests of these modes (tests/test_gpu_arith16.py), with the RMS deviation bounded four times tighter; the pin that
    DISCRIMINATES is the tiny architecture (two flows of two layers: the operand rounding is matched before the noise builds up — bf16 z_flow agrees
    to 2e-7 where the fp32 oracle is 2e-3 away) and the operator-level fixtures below."""
    g = golden("%s_synth_arith_%s_taps.npz" % (arch, name))
    f32 = golden("%s_synth_refmode_taps.npz" % arch)
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY if arch == "tiny" else pkg.SYNTH_FULL))
    arith = oracle.ARITH_F16 if name == "f16" else oracle.ARITH_BF16
    kw = dict(mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_EXPLICIT, noise_dur=g["noise_dur"], noise_prior=g["noise_prior"])
    r = m.process_ids(g["ids"], arith=arith, **kw)
    np.testing.assert_array_equal(r["durations"], g["durations"].ravel())
    np.testing.assert_array_equal(g["durations"], f32["durations"])  # the arithmetic mode does not move the durations (default scope)
    for tap in ("enc_out", "prior_mean", "prior_logvar", "log_duration", "z_p"):
        assert rel_err(r[tap], g[tap]) < 1e-4, tap

    def dev(a, b):
        a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
        rms = np.sqrt((b ** 2).mean())
        return np.abs(a - b).max() / rms, np.sqrt(((a - b) ** 2).mean()) / rms

    for tap in ("z_flow", "pre_tanh", "waveform"):
        mx, rms = dev(r[tap], g[tap])
        assert mx < tol and rms < tol / 4, (tap, mx, rms)
    # the fixture really is in another arithmetic than fp32: it differs from the fp32 reference-mode fixture by the mode's rounding noise
    assert dev(g["waveform"], f32["waveform"])[0] > (5e-4 if name == "f16" else 4e-3)
    if arch == "tiny":
        # discrimination: the oracle in fp32 arithmetic is several times further from the fixture's flow output than the oracle in the fixture's arithmetic
        r32 = m.process_ids(g["ids"], taps=["z_flow"], **kw)
        assert dev(r["z_flow"], g["z_flow"])[1] * 3 < dev(r32["z_flow"], g["z_flow"])[1]
        assert dev(r["z_flow"], g["z_flow"])[1] < (1e-4 if name == "f16" else 1e-5)


def test_oracle_16bit_conv_operators_reproduce_torch_with_rounded_operands(oracle):
    """Operator-level third-party pin of the 16-bit-operand arithmetic: tests/golden/arith16_ops.npz holds inputs and torch outputs of
    Conv1d (taps 1 / 3 / 5 / 7 / 11, dilations 1 / 3 / 5, fused input LeakyReLU) and ConvTranspose1d (strides 8 / 2, reference crop 0 and HF crop)
    with BOTH operands rounded to fp16 / bf16 and fp32 accumulation (make_golden.py `arith16_op_fixtures`). One conv differs between two
    correct implementations by fp32 summation order only: 2e-5 of RMS, the fp32 bound — far below the modes' own rounding step, which the oracle
    WITHOUT the operand rounding (arith = fp32) must therefore miss."""
    g = golden("arith16_ops.npz")
    n = int(g["n_cases"][0])
    assert n >= 8
    for i in range(n):
        meta = g["meta_%d" % i]  # kind (0 conv, 1 transposed), dilation | stride, crop, slope x 1e6
        kind, p1, crop, slope = int(meta[0]), int(meta[1]), int(meta[2]), float(meta[3]) / 1e6
        x, w, b = g["x_%d" % i], g["w_%d" % i], g["b_%d" % i]
        for name, arith in (("f16", oracle.ARITH_F16), ("bf16", oracle.ARITH_BF16)):
            y = g["y_%s_%d" % (name, i)]
            if kind == 0:
                extra = dict(pre_slope=slope) if slope != 1.0 else {}
                got = oracle.conv1d(x, w, b, dilation=p1, arith=arith, **extra)
                plain = oracle.conv1d(x, w, b, dilation=p1, arith=0, **extra)
            else:
                got = oracle.conv_transpose1d(x, w, b, p1, crop, pre_slope=slope, arith=arith)
                plain = oracle.conv_transpose1d(x, w, b, p1, crop, pre_slope=slope, arith=0)
            assert got.shape == y.shape, (i, got.shape, y.shape)
            assert rel_err(got, y) < 2e-5, (i, name, meta.tolist(), rel_err(got, y))
            assert rel_err(plain, y) > (2e-4 if name == "f16" else 1.5e-3), (i, name, "fp32 arithmetic is indistinguishable here", rel_err(plain, y))


def test_exact_math_header_conversions_and_polynomials(tmp_path):
    """include/vits_exact_math.h, the arithmetic the emulated-ggml mode shares between device and oracle: its integer fp32 <-> fp16 conversions
    against the CPU's own (F16C, round to nearest even) on every half and on 2^26 floats incl. the subnormal / overflow / tie ranges; its
    polynomial exp / log / softplus within a few ulp of the C library (they need not be better: both sides use THESE, what matters is that they
    are the functions they claim to be)."""
    import subprocess
    src = tmp_path / "vx.cpp"
    src.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <immintrin.h>
#include "vits_exact_math.h"
static float ulps(float a, float b) { return std::fabs(a - b) / std::fabs(std::nextafter(b, INFINITY) - b); }
int main() {
    long bad = 0;
    for (uint32_t h = 0; h < 65536; ++h) {
        const float f = vx_f16_to_f32((uint16_t)h), g = _cvtsh_ss((uint16_t)h);
        if (std::memcmp(&f, &g, 4) != 0 && !(f != f && g != g)) ++bad;
        if (f == f && vx_f32_to_f16(f) != (uint16_t)h) ++bad;
    }
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < (1l << 26); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t u = (uint32_t)s;
        if (i & 1) u = (u & 0x807fffffu) | ((uint32_t)(96 + (s >> 40) % 48) << 23);  // exponents around the fp16 range, subnormals and overflow included
        if ((i & 15) == 3) u &= 0xffffe000u | 0x1000u;                                  // exact ties
        float f; std::memcpy(&f, &u, 4);
        if (f != f) continue;
        if (vx_f32_to_f16(f) != (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT)) ++bad;
    }
    float worst_e = 0, worst_l = 0, worst_s = 0;
    for (int i = 0; i <= 2000000; ++i) {
        const float x = -87.0f + 175.0f * (float)i / 2000000.0f;
        worst_e = std::fmax(worst_e, ulps(vx_expf(x), (float)std::exp((double)x)));
        const float y = std::exp(-20.0f + 40.0f * (float)i / 2000000.0f);
        worst_l = std::fmax(worst_l, std::fabs(vx_logf(y) - (float)std::log((double)y)) / std::fmax(1.2e-7f * std::fabs((float)std::log((double)y)), 1.2e-7f));
        const float z = -30.0f + 55.0f * (float)i / 2000000.0f;
        const float want = z > 20.0f ? z : (float)std::log1p(std::exp((double)z));
        worst_s = std::fmax(worst_s, ulps(vx_softplusf(z), want));
    }
    std::printf("%ld %g %g %g %g\n", bad, worst_e, worst_l, worst_s, vx_duration(0.0f, 1.0f) + vx_duration(1.0f, 1.0f) * 10 + vx_duration(-5.0f, 1.0f) * 100);
    return 0;
}
''')
    exe = tmp_path / "vx"
    cc = subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include"),
                         str(src), "-o", str(exe)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    bad, we, wl, ws, dsum = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300).stdout.split()
    assert int(bad) == 0
    assert float(we) < 4 and float(wl) < 8 and float(ws) < 8, (we, wl, ws)
    assert float(dsum) == 1 + 3 * 10 + 1 * 100  # ceil(exp(0)) = 1, ceil(e) = 3, ceil(exp(-5)) = 1


def test_oracle_exact_order_table_mode_agrees_with_its_independent_loops(pkg, oracle):
    """vo_opts.ggml_tables = 1 (stage one in the shared exact order, oracle/vits_oracle_exact.cpp) against = 2 (THIS oracle's own loops — another
    summation grouping, the C library's exp / log — with the same table lookups): two restatements of the same lines, so the log-durations agree at
    the size of the tables' own rounding noise (a lookup flips on one side where its argument sits on an fp16 rounding boundary: bound 5e-2 of RMS,
    measured ~5e-3) and the durations on all but a handful of ids; both move away from the table-free arithmetic by the same amount. Both modes."""
    m = oracle.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
    ids = pkg.synth_ids(6, 96)
    differ = total = 0
    for u in range(6):
        for mode in (oracle.MODE_REFERENCE, oracle.MODE_HF):
            l0, d0 = m.log_durations(ids[u], mode=mode, noise_seed=4321 + u)
            l1, d1 = m.log_durations(ids[u], mode=mode, noise_seed=4321 + u, ggml_tables=1)
            l2, d2 = m.log_durations(ids[u], mode=mode, noise_seed=4321 + u, ggml_tables=2)
            assert rel_err(l1, l2) < 5e-2 and rel_err(l1, l0) < 5e-2
            assert np.abs(l1 - l0).max() > 1e-6  # the tables are on
            differ += int((d1 != d2).sum())
            total += d1.size
    assert differ <= max(3, total // 200), (differ, total)
