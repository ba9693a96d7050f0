"""16-bit-operand arithmetic modes (VITS_ARITH_F16 = the reference's own conv arithmetic, SURVEY.md App. B Q7:
fp16 im2col x fp16 weights -> fp32, /root/reference/src/include/custom-ops.h:684-690; VITS_ARITH_BF16 = BASELINE.json
configs[4]) on v_mfma_f32_32x32x16_{f16,bf16}, against the CPU oracle with the SAME operand rounding (vo_opts.arith).

Tolerances: both sides round the same fp32 values to the same 16-bit operands and the products are exact in fp32, so one
conv differs only by fp32 summation order (<= 2e-5 of RMS, as in fp32 mode). Across a whole model an fp32 value that lands
within an ulp of a rounding boundary can round the other way on the two sides (1 unit in the last place of a 16-bit number =
1e-3 / 8e-3 relative, on isolated elements), so whole-model FLOAT taps are compared at 5e-3 (fp16) / 8e-2 (bf16) of RMS — the
size of the mode's own rounding noise ("report only" against fp32).

Integer outputs carry NO tolerance. Under the default scope VITS_ARITH_SCOPE_FLOW_VOCODER stage one (text encoder + duration
predictor) stays exact fp32, so durations / frame counts / sample counts must equal the oracle's — and the fp32 path's — bit
for bit in every arithmetic mode (/root/reference/src/vits.cpp:996-1001). Under VITS_ARITH_SCOPE_ALL_CONVS (the literal Q7:
the duration predictor's convs round their inputs too) the log-durations are a float tap like any other and are compared as
one; the integer durations of that scope are a function of a value that carries rounding noise and are not a parity target."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
ARITHS = [("f16", 2, 5e-3), ("bf16", 1, 8e-2)]


@pytest.fixture(autouse=True)
def _reset_op_arith(pkg):
    yield
    pkg.op_set_arith(pkg.ARITH_F32)


SHAPES = [
    # (cin, cout, k, dil, T, B)
    (192, 512, 7, 1, 200, 2), (256, 256, 11, 1, 300, 2), (256, 256, 11, 5, 300, 1), (128, 128, 7, 3, 700, 2), (64, 64, 3, 5, 900, 1),
    (32, 32, 11, 3, 1500, 1), (32, 32, 3, 1, 257, 3), (192, 768, 3, 1, 40, 4), (768, 192, 3, 1, 40, 4), (96, 192, 1, 1, 130, 2),
    (192, 29, 1, 1, 33, 2), (16, 24, 5, 2, 50, 2), (8, 16, 3, 3, 19, 1), (20, 12, 5, 1, 64, 2),
]


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
@pytest.mark.parametrize("cin,cout,k,dil,T,B", SHAPES)
def test_conv1d_16bit_matches_oracle(pkg, oracle, name, arith, _tol, cin, cout, k, dil, T, B):
    rng = np.random.default_rng(cin * 131 + k * 7 + dil)
    x = rng.standard_normal((B, cin, T)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    lens = np.array([T] + [max(1, T - 7 * (b + 1)) for b in range(B - 1)], np.int32)
    res = rng.standard_normal((B, cout, T)).astype(np.float32)
    pkg.op_set_arith(arith)
    for kw in (dict(), dict(pre_slope=0.1, residual=res), dict(post_act=1, accum=res, out_scale=1.0 / 3)):
        got = pkg.op_conv1d(x, w, bias, dilation=dil, lens=lens, **kw)
        want = oracle.conv1d(x, w, bias, dilation=dil, lens=lens, arith=arith, **kw)
        for b in range(B):
            assert rel_err(got[b, :, : lens[b]], want[b, :, : lens[b]]) < 2e-5, (name, kw.keys(), b)


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
def test_gated_conv_and_conv_transpose_16bit_match_oracle(pkg, oracle, name, arith, _tol):
    rng = np.random.default_rng(9)
    pkg.op_set_arith(arith)
    x = rng.standard_normal((2, 192, 150)).astype(np.float32)
    w = (rng.standard_normal((384, 192, 5)) / np.sqrt(192 * 5)).astype(np.float32)
    bias = rng.standard_normal(384).astype(np.float32)
    lens = np.array([150, 77], np.int32)
    got = pkg.op_conv1d(x, w, bias, post_act=2, lens=lens)
    want = oracle.conv1d(x, w, bias, post_act=2, lens=lens, arith=arith)
    for b in range(2):
        assert rel_err(got[b, :, : lens[b]], want[b, :, : lens[b]]) < 2e-5
    for cin, cout, k, s, T in ((512, 256, 16, 8, 60), (128, 64, 4, 2, 333), (32, 16, 8, 4, 21)):
        x = rng.standard_normal((2, cin, T)).astype(np.float32)
        w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * 2)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        lens = np.array([T, T - 5], np.int32)
        for crop in (0, (k - s) // 2):
            got = pkg.op_conv_transpose1d(x, w, bias, s, crop, pre_slope=0.1, lens=lens)
            want = oracle.conv_transpose1d(x, w, bias, s, crop, pre_slope=0.1, lens=lens, arith=arith)
            for b in range(2):
                n = s * lens[b] + k - s - 2 * crop
                assert rel_err(got[b, :, :n], want[b, :, :n]) < 2e-5, (cin, crop, b)


def _ids(T, seed, vocab=38):
    rng = np.random.default_rng(seed)
    ids = np.zeros(T, np.int32)
    ids[1::2] = rng.integers(1, vocab, size=len(ids[1::2]))
    return ids


@pytest.mark.parametrize("name,arith,tol", ARITHS)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("group", [True, False])
@pytest.mark.parametrize("scope", [0, 1], ids=["flow_vocoder", "all_convs"])
def test_full_model_16bit_matches_oracle(pkg, oracle, full_bytes, monkeypatch, name, arith, tol, mode, group, scope):
    """Whole path in a 16-bit mode vs the oracle in the same mode and scope, with the vocoder in the group layout (default) and
    through the transparent fp32-layout path (VITS_NO_GROUP16=1): both must agree with the oracle."""
    if not group:
        monkeypatch.setenv("VITS_NO_GROUP16", "1")
    ids = _ids(33, 5)
    om = oracle.Model(full_bytes)
    with pkg.Model(full_bytes) as m:
        assert m.arith_scope == pkg.SCOPE_FLOW_VOCODER  # the default
        m.set_arith_scope(scope)
        m.set_arith(arith)
        assert m.arith == arith and m.arith_scope == scope
        # durations pinned: every float tap is then compared on identical shapes
        pcm, lengths, frames = m.process_batch(ids, mode=mode, noise_seed=11, fixed_duration=2, collect_taps=True)
        ref = om.process_ids(ids, mode=mode, noise_kind=oracle.NOISE_COUNTER, noise_seed=11, fixed_duration=2, arith=arith, arith_scope=scope)
        assert lengths[0] == ref["waveform"].size
        for tap in ("enc_out", "prior_mean", "log_duration", "z_p", "z_flow", "pre_tanh", "waveform"):
            # (stage one is exact fp32 under the flow_vocoder scope: its taps meet the fp32 bound there)
            bound = 1e-4 if (scope == 0 and tap in ("enc_out", "prior_mean", "log_duration", "z_p")) else tol
            assert rel_err(m.tap(tap), ref[tap]) < bound, (name, tap)
        if scope == 0:
            # predicted durations: bit-exact against the oracle in the same arithmetic, no tolerance, and the waveform
            # unconditionally on the same shapes
            pcm, lengths, frames = m.process_batch(ids, mode=mode, noise_seed=11, collect_taps=True)
            ref = om.process_ids(ids, mode=mode, noise_kind=oracle.NOISE_COUNTER, noise_seed=11, arith=arith, arith_scope=scope)
            np.testing.assert_array_equal(m.tap("durations"), ref["durations"])
            assert frames[0] == int(ref["durations"].sum()) and lengths[0] == ref["waveform"].size
            assert rel_err(pcm[0], ref["waveform"]) < tol


@pytest.mark.parametrize("name,arith,tol", ARITHS)
def test_durations_do_not_depend_on_the_arithmetic_mode(pkg, full_bytes, name, arith, tol):
    """VITS_ARITH_SCOPE_FLOW_VOCODER: stage one runs the SAME fp32 kernels in every arithmetic mode — log-durations, durations, frame
    and sample counts of a ragged batch are bit-identical to the fp32 path's; the all-convs scope is the one that moves them."""
    Ts = [40, 7, 33, 1]
    ids = np.zeros((4, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 20 + b)
    with pkg.Model(full_bytes) as m:
        base = m.process_batch(ids, id_lengths=Ts, noise_seed=5, collect_taps=True)
        base_logw = [m.tap("log_duration", u).copy() for u in range(4)]
        base_dur = [m.tap("durations", u).copy() for u in range(4)]
        m.set_arith(arith)
        got = m.process_batch(ids, id_lengths=Ts, noise_seed=5, collect_taps=True)
        assert np.array_equal(got[1], base[1]) and np.array_equal(got[2], base[2])
        for u in range(4):
            assert np.array_equal(m.tap("log_duration", u), base_logw[u]) and np.array_equal(m.tap("durations", u), base_dur[u])
            assert 1e-6 < rel_err(got[0][u], base[0][u]) < 4 * tol  # the audio does change: the flow and the vocoder are 16-bit
        m.set_arith_scope(pkg.SCOPE_ALL_CONVS)
        m.process_batch(ids, id_lengths=Ts, noise_seed=5, collect_taps=True)
        assert not np.array_equal(m.tap("log_duration", 0), base_logw[0])
        assert rel_err(m.tap("log_duration", 0), base_logw[0]) < 4 * tol  # (against fp32, not against the oracle in the same arithmetic)


@pytest.mark.parametrize("name,arith,tol", ARITHS)
def test_benchmark_batch_durations_are_bit_exact_in_16bit_modes(pkg, oracle, full_bytes, name, arith, tol):
    """The bench's own batch (64 x 128 ids, ids seed 1234+u, noise seed 4321+u, reference mode, predicted durations) in f16 / bf16
    arithmetic: four utterances against the oracle in the same arithmetic — all 128 durations each, frame and sample counts exact;
    waveform at the mode's noise level."""
    ids = pkg.synth_ids(64, 128)
    om = oracle.Model(full_bytes)
    with pkg.Model(full_bytes) as m:
        m.set_arith(arith)
        pcm, lengths, frames = m.process_batch(ids, noise_seed=4321, collect_taps=True)
        for u in (0, 21, 42, 63):
            ref = om.process_ids(ids[u], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_COUNTER, noise_seed=4321 + u, arith=arith)
            np.testing.assert_array_equal(m.tap("durations", u), ref["durations"])
            assert frames[u] == int(ref["durations"].sum()) and lengths[u] == ref["waveform"].size == 256 * frames[u] + 294
            assert rel_err(m.tap("z_flow", u), ref["z_flow"]) < tol
            assert rel_err(pcm[u], ref["waveform"]) < 2 * tol, (u, rel_err(pcm[u], ref["waveform"]))


def test_config5_in_its_own_arithmetic(pkg, oracle):
    """BASELINE.json configs[4] at its own size AND precision: the two bf16-stored models (seeds 0x5EED / 0xBEEF), 1024-id utterances,
    VITS_ARITH_BF16, predicted durations, against oracle(arith = bf16): durations / frames / sample counts bit-exact, the waveform
    within the bf16 rounding-noise bound (maximum and RMS deviation), windowed vocoder == whole utterance bit for bit."""
    data = [pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL | pkg.SYNTH_BF16), pkg.synth_model_bytes(0xBEEF, pkg.SYNTH_FULL | pkg.SYNTH_BF16)]
    ids = [pkg.synth_ids(2, 1024, ids_seed=1234), pkg.synth_ids(2, 1024, ids_seed=91234)]
    for k in (0, 1):
        with pkg.Model(data[k]) as m:
            m.set_arith(pkg.ARITH_BF16)
            pcm, lengths, frames = m.process_batch(ids[k], noise_seed=50 + k, collect_taps=True)
            assert (frames > 1024).all()
            ref = oracle.Model(data[k]).process_ids(ids[k][0], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_COUNTER, noise_seed=50 + k,
                                                    arith=oracle.ARITH_BF16, taps=["durations", "waveform"])
            np.testing.assert_array_equal(m.tap("durations", 0), ref["durations"])
            assert frames[0] == int(ref["durations"].sum()) and lengths[0] == ref["waveform"].size == 256 * frames[0] + 294
            # half a million samples: the maximum sits at the tail of the rounding-flip distribution; bound it and the RMS deviation
            d = pcm[0].astype(np.float64) - ref["waveform"]
            rms = np.sqrt((ref["waveform"].astype(np.float64) ** 2).mean())
            print("config 5 model %d bf16: max |d| / RMS = %.3e, RMS(d) / RMS = %.3e" % (k, np.abs(d).max() / rms, np.sqrt((d ** 2).mean()) / rms))
            # (measured: max 3.5e-2 / 4.6e-2, RMS 6.9e-3 / 7.7e-3 on the two models; 8 x the fp16 figures of the test below, as the mantissas)
            assert np.abs(d).max() / rms < 0.1 and np.sqrt((d ** 2).mean()) / rms < 1.5e-2
            tiled, lt, _ = m.process_batch(ids[k], noise_seed=50 + k, vocoder_chunk_frames=256)
            assert np.array_equal(lt, lengths) and all(np.array_equal(a, b) for a, b in zip(tiled, pcm))


def test_f16_mode_is_the_reference_arithmetic_report_only(pkg, full_bytes):
    """"ref_quirks + fp16-activation emulation vs fp32: report only (expected 1e-3 ... 1e-2)". The f16 mode
    (Q7) moves the waveform by about 1e-3 of its RMS against exact fp32, bf16 by about 1e-2; neither is bit-equal to fp32."""
    ids = _ids(40, 3)
    with pkg.Model(full_bytes) as m:
        base, _, _ = m.process_batch(ids, noise_seed=2, fixed_duration=2)
        deltas = {}
        for name, arith, _ in ARITHS:
            m.set_arith(arith)
            out, _, _ = m.process_batch(ids, noise_seed=2, fixed_duration=2)
            deltas[name] = rel_err(out[0], base[0])
        m.set_arith(pkg.ARITH_F32)
        again, _, _ = m.process_batch(ids, noise_seed=2, fixed_duration=2)
        assert np.array_equal(again[0], base[0])  # switching back restores the exact path bit for bit
    print("16-bit modes vs fp32 (max |d| / RMS):", deltas)
    assert 1e-5 < deltas["f16"] < 2e-2 and 1e-4 < deltas["bf16"] < 2e-1


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
def test_16bit_batch_and_window_invariance(pkg, full_bytes, name, arith, _tol):
    """Utterances never interact and window edges see exact halos, in the 16-bit modes too: a ragged batch equals batch-1 runs
    and the windowed vocoder equals the whole-utterance run, bit for bit (the k-order of an output's products does not depend
    on the tile, the batch or the window)."""
    Ts = [24, 9, 40]
    ids = np.zeros((3, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 60 + b)
    with pkg.Model(full_bytes) as m:
        m.set_arith(arith)
        pcm, lengths, frames = m.process_batch(ids, id_lengths=Ts, noise_seed=70)
        for b, T in enumerate(Ts):
            one, l1, _ = m.process_batch(ids[b:b + 1, :T], noise_seed=70 + b)
            assert l1[0] == lengths[b] and np.array_equal(one[0], pcm[b]), b
        tiled, lt, _ = m.process_batch(ids, id_lengths=Ts, noise_seed=70, vocoder_chunk_frames=16)
        assert np.array_equal(lt, lengths) and int(frames.max()) > 16
        for a, b_ in zip(pcm, tiled):
            assert np.array_equal(a, b_)


@pytest.mark.parametrize("fixture", ["tiny_hf", "tiny_synth"])
def test_tiny_models_run_in_16bit_modes(pkg, oracle, tiny_bytes, tiny_hf_bytes, fixture):
    """Other kernel sizes / dilations / channel counts than the MMS-TTS architecture (run-time-dilation kernels, channel counts
    below 32): f16 mode against the oracle."""
    data = tiny_hf_bytes if fixture == "tiny_hf" else tiny_bytes
    om = oracle.Model(data)
    ids = _ids(21, 8)
    with pkg.Model(data) as m:
        m.set_arith(pkg.ARITH_F16)
        m.set_arith_scope(pkg.SCOPE_ALL_CONVS)  # (the encoder's convs in fp16 as well: the literal Q7)
        pcm, lengths, _ = m.process_batch(ids, noise_seed=4, fixed_duration=3, collect_taps=True)
        ref = om.process_ids(ids, noise_kind=oracle.NOISE_COUNTER, noise_seed=4, fixed_duration=3, arith=oracle.ARITH_F16, arith_scope=oracle.SCOPE_ALL_CONVS)
        assert lengths[0] == ref["waveform"].size
        for tap in ("enc_out", "z_flow", "pre_tanh"):
            assert rel_err(m.tap(tap), ref[tap]) < 3e-3, tap


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
def test_fused_resblock_pair_is_bit_identical_to_the_two_kernel_path(pkg, full_bytes, monkeypatch, name, arith, _tol):
    """rbpair16.hip (C = 32 / 64: conv1 -> LDS -> conv2 in one kernel) uses the same operands, rounding points and k-order as two
    conv16 launches: the PCM must not move by a bit — ragged batch, windowed vocoder, both semantics modes."""
    Ts = [30, 11, 40]
    ids = np.zeros((3, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 90 + b)
    outs = {}
    long_ids = pkg.synth_ids(2, 700, ids_seed=4242)  # 1400 frames: several times more blocks per launch than the GPU holds at once
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("VITS_NO_FUSE16", "1")
        with pkg.Model(full_bytes) as m:
            m.set_arith(arith)
            for mode in (0, 1):
                outs[(fused, mode, 0)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33)
                outs[(fused, mode, 1)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33, vocoder_chunk_frames=24)
            outs[(fused, "long", 0)] = m.process_batch(long_ids, noise_seed=34, fixed_duration=2)
    # long utterances: blocks of one launch start after others have finished — what exposes a block reading data another block of the
    # same launch has already overwritten (the fused pairs must not write the 16-bit stream they read)
    for x, y in zip(outs[(True, "long", 0)][0], outs[(False, "long", 0)][0]):
        assert np.array_equal(x, y), "long utterance"
    for mode in (0, 1):
        for w in (0, 1):
            a, b_ = outs[(True, mode, w)], outs[(False, mode, w)]
            assert np.array_equal(a[1], b_[1])
            for x, y in zip(a[0], b_[0]):
                assert np.array_equal(x, y), (mode, w)
        for x, y in zip(outs[(True, mode, 0)][0], outs[(True, mode, 1)][0]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
def test_whole_resblock_kernel_is_bit_identical_to_the_pair_path(pkg, full_bytes, monkeypatch, name, arith, _tol):
    """rbblock16.hip (C = 32 / 64: the three conv pairs of a resblock as ONE kernel, the fp32 stream in registers, x / t in one LDS tile)
    against three rbpair16 launches (VITS_NO_RBBLOCK16=1): same operands, rounding points and k-order, so the PCM must not move by a bit —
    ragged batch with very short members (tiles that are mostly sequence-end padding), windowed vocoder, both semantics modes, and
    utterances long enough that a launch runs many rounds of blocks (/root/reference/src/vits.cpp:545-581,622-635)."""
    Ts = [30, 11, 40, 1, 2]
    ids = np.zeros((5, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 90 + b)
    long_ids = pkg.synth_ids(2, 700, ids_seed=4242)
    outs = {}
    for block in (True, False):
        if not block:
            monkeypatch.setenv("VITS_NO_RBBLOCK16", "1")
        with pkg.Model(full_bytes) as m:
            m.set_arith(arith)
            for mode in (0, 1):
                outs[(block, mode, 0)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33)
                outs[(block, mode, 1)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33, vocoder_chunk_frames=24)
            outs[(block, "long", 0)] = m.process_batch(long_ids, noise_seed=34, fixed_duration=2)
            m.prof_enable(True)
            outs[(block, "prof", 0)] = m.process_batch(ids, id_lengths=Ts, noise_seed=33)
            names = [k["name"] for k in m.prof_report()["kernels"]]
            assert any("hifigan_resblock_block" in n for n in names) == block, names
            m.prof_enable(False)
    for x, y in zip(outs[(True, "long", 0)][0], outs[(False, "long", 0)][0]):
        assert np.array_equal(x, y), "long utterance"
    for x, y in zip(outs[(True, "prof", 0)][0], outs[(True, 0, 0)][0]):
        assert np.array_equal(x, y), "profiler (serial) run"
    for mode in (0, 1):
        for w in (0, 1):
            a, b_ = outs[(True, mode, w)], outs[(False, mode, w)]
            assert np.array_equal(a[1], b_[1])
            for x, y in zip(a[0], b_[0]):
                assert np.array_equal(x, y), (mode, w)
        for x, y in zip(outs[(True, mode, 0)][0], outs[(True, mode, 1)][0]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["f16", "bf16"])
def test_16bit_kernel_choices_do_not_change_a_single_bit(name):
    """The 16-bit path has several interchangeable kernels per layer — streaming transposed conv (convt16.hip) or GEMM tile, whole-resblock
    kernel or fused pairs or two launches per pair, group layout or converter path: same operands, rounding points and k-order, so the PCM
    of ragged batches (short and 300-id utterances, whole and windowed, both semantics modes) must hash identically whichever runs. One
    process per setting (the kernel files read their knobs once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "knob_identity.py")

    def run(extra):
        env = dict(os.environ)
        env.update(extra)
        env["VITS_KNOB_ARITH"] = name
        out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.strip().splitlines()[-1]

    base = run({})
    for extra in ({"VITS_NO_CONVT16S": "1"}, {"VITS_NO_CONVT16L": "1"}, {"VITS_CONVT16S_ALL": "1"}, {"VITS_NO_RBBLOCK16": "1"}, {"VITS_RBB_C64K11": "1"}, {"VITS_RBB_C128": "0"},
                  {"VITS_NO_FUSE16": "1", "VITS_NO_CONVT16S": "1"}, {"VITS_RB_STREAMS": "1"}, {"VITS_NO_FLOW_FUSE": "1"}, {"VITS_FLOW_NCW": "1"},
                  {"VITS_NO_FLOW_FUSE": "1", "VITS_NO_WN_FUSE": "1"}, {"VITS_NO_FLOW_FUSE": "1", "VITS_WN16_NCW": "2"}, {"VITS_ATT_NW": "8"}, {"VITS_ATT_SHORT": "0"}, {"VITS_NO_LAT16": "1"}, {"VITS_FLOW_NARROW_MAX": "0"}, {"VITS_FLOW_NARROW_MAX": "100000"}, {"VITS_RB16_NARROW_MAX": "0"}, {"VITS_RB16_NARROW_MAX": "100000"}, {"VITS_CONVT16_SPLIT_MAX": "0"}, {"VITS_CONVT16_SPLIT_MAX": "100000"}, {"VITS_RB16_SERIAL_MAX_FRAMES": "0"}, {"VITS_RB16_SERIAL_MAX_FRAMES": "1000000"}, {"VITS_RB16_SERIAL_MIN_FRAMES": "0"}, {"VITS_RB16_SERIAL_MIN_FRAMES": "1000000"}, {"VITS_KEEP_STAGE_SUM32": "1"},
                  {"VITS_NO_RB_SUM3": "1"}, {"VITS_RB16_SERIAL_MIN_FRAMES": "1000000", "VITS_RB16_NARROW_MAX": "0"}, {"VITS_NO_DDS_LAT": "1"}, {"VITS_NO_LN_FUSE": "1"}, {"VITS_NO_DDS_LAT": "1", "VITS_NO_LN_FUSE": "1"}, {"VITS_DDS_LAT_MAX_BLOCKS": "2"},
                  {"VITS_KEEP_STAGE_SUM32": "1", "VITS_NO_RBBLOCK16": "1"}, {"VITS_NO_FUSE16": "1"},
                  # whole-resblock kernels walking segments of 2 / 3 / 8 tiles with the left halo taken from the previous tile (default: per shape and only
                  # on grids of thousands of blocks — forced here on every shape and from two tiles up), and never
                  {"VITS_RBB_STREAM_MIN_BLOCKS": "1", "VITS_RBB_STREAM_TILES": "2"}, {"VITS_RBB_STREAM_MIN_BLOCKS": "1", "VITS_RBB_STREAM_TILES": "3"},
                  {"VITS_RBB_STREAM_MIN_BLOCKS": "1", "VITS_RBB_STREAM_TILES": "8"}, {"VITS_RBB_STREAM_MIN_BLOCKS": "1", "VITS_RBB_STREAM_TILES": "5", "VITS_RBB_C64K11": "1"},
                  {"VITS_RBB_STREAM_MIN_BLOCKS": "1"}, {"VITS_RBB_STREAM_TILES": "0"},
                  # the wide stages' resblock convs on small grids: conv16_lat_kernel (default up to 2048 tiles at C = 256) in its three block shapes, on every
                  # grid and at C = 128 as well, never; the side-by-side resblock sum only for whole-resblock stages / in the old enqueue order
                  {"VITS_NO_RBB_GROUP3": "1"}, {"VITS_NO_RBB_GROUP3_C64": "1"}, {"VITS_NO_RBB_GROUP3": "1", "VITS_NO_LAT16H_GROUP": "1"},
                  {"VITS_NO_LAT16H": "1"}, {"VITS_NO_LAT16H_PRE": "1"}, {"VITS_NO_LAT16H_GROUP": "1"}, {"VITS_LAT16H_GROUP_SHAPE": "42"}, {"VITS_LAT16H_GROUP_SHAPE": "22", "VITS_LAT16H_MAX_TILES": "1000000"}, {"VITS_LAT16H_SHAPE": "22"}, {"VITS_LAT16H_SHAPE": "42"}, {"VITS_LAT16H_MAX_TILES": "1000000", "VITS_LAT16H_MAX_TILES_C128": "1000000"},
                  {"VITS_LAT16H_MAX_TILES": "1000000", "VITS_LAT16H_MAX_TILES_C128": "1000000", "VITS_LAT16H_SHAPE": "42", "VITS_RB16_SERIAL_MIN_FRAMES": "1000000"},
                  {"VITS_NO_ATT_LAT": "1"}, {"VITS_RB_SUM3_BLOCK_ONLY": "1"}, {"VITS_RB_SUM3_IN_ORDER": "1"}, {"VITS_NO_LAT16H": "1", "VITS_RB_SUM3_BLOCK_ONLY": "1", "VITS_RB_SUM3_IN_ORDER": "1"},
                  # the flow as two chains of launches over halves of the batch (default only above 256 blocks per layer: forced here), and never
                  {"VITS_FLOW_CHAIN_MIN_BLOCKS": "0", "VITS_FLOW_NARROW_MAX": "0"}, {"VITS_FLOW_CHAIN_MIN_BLOCKS": "0", "VITS_FLOW_NARROW_MAX": "2"}, {"VITS_FLOW_CHAINS": "1"}):
        assert run(extra) == base, extra


def test_long_form_1024_ids_in_f16_mode(pkg, oracle, full_bytes):
    """BASELINE config 5 input length in the reference's arithmetic: a 1024-id utterance (pinned durations: 2048 frames, 33 s of
    audio) against the oracle in fp16 mode, the windowed vocoder bit-identical to the whole-utterance run, and the bf16 mode
    bit-identical between windowed and whole as well (long sequences: several hundred column tiles per stage, fused resblock
    pairs across many blocks)."""
    ids = pkg.synth_ids(1, 1024, ids_seed=77)[0]
    om = oracle.Model(full_bytes)
    with pkg.Model(full_bytes) as m:
        m.set_arith(pkg.ARITH_F16)
        whole, lw, fw = m.process_batch(ids, noise_seed=12, fixed_duration=2)
        assert fw[0] == 2048
        ref = om.process_ids(ids, noise_kind=oracle.NOISE_COUNTER, noise_seed=12, fixed_duration=2, arith=oracle.ARITH_F16, taps=["waveform"])["waveform"]
        assert lw[0] == ref.size
        # over half a million samples the MAXIMUM deviation sits at the tail of the rounding-flip distribution (the oracle's own
        # fp16-vs-fp32 maximum on this utterance is 3.8e-3): bound the maximum at 1e-2 and the RMS deviation at 2e-3 (measured 1.0e-3).
        # A fused block that read its neighbour's already-overwritten halo (in-place 16-bit stream) produced maxima of 4e-2 to 7e-2
        # on ~200 samples here while every short-utterance test passed: this utterance is long enough to expose block-order races.
        assert rel_err(whole[0], ref) < 1e-2
        d = whole[0].astype(np.float64) - ref
        assert np.sqrt((d ** 2).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean()) < 2e-3
        tiled, lt, _ = m.process_batch(ids, noise_seed=12, fixed_duration=2, vocoder_chunk_frames=256)
        assert np.array_equal(tiled[0], whole[0])
        m.set_arith(pkg.ARITH_BF16)
        whole_b, _, _ = m.process_batch(ids, noise_seed=12, fixed_duration=2)
        tiled_b, _, _ = m.process_batch(ids, noise_seed=12, fixed_duration=2, vocoder_chunk_frames=300)
        assert np.array_equal(tiled_b[0], whole_b[0])
        assert 1e-4 < rel_err(whole_b[0], whole[0]) < 0.3


@pytest.mark.parametrize("name,arith", [("f32", 0)] + [(n, a) for n, a, _ in ARITHS])
def test_fused_dds_layer_is_bit_identical_to_the_three_kernel_path(pkg, full_bytes, monkeypatch, name, arith):
    """misc_kernels.hip dds_layer_kernel (depthwise + LN + gelu + 1x1 conv + LN + gelu + residual of one DDS layer, vits.cpp:655-691,
    as one kernel) takes every sum in the order of the three launches it replaces: the log-durations and everything behind them must
    not move by a bit — ragged batch with very short and long members (more 32-column tiles than one launch runs at once), both
    semantics modes, all arithmetic modes."""
    Ts = [30, 1, 40, 3, 33]
    ids = np.zeros((5, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 70 + b)
    long_ids = pkg.synth_ids(3, 900, ids_seed=77)
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("VITS_NO_DDS_FUSE", "1")
        with pkg.Model(full_bytes) as m:
            m.set_arith(arith)
            m.set_arith_scope(pkg.SCOPE_ALL_CONVS)  # (so that the 16-bit variants of the DDS kernels are the ones compared)
            for mode in (0, 1):
                pcm, lengths, _ = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=12, collect_taps=True)
                outs[(fused, mode)] = (pcm, lengths, m.tap("log_duration").copy())
            outs[(fused, "long")] = m.process_batch(long_ids, noise_seed=13, frames_only=True)
    assert np.array_equal(outs[(True, "long")][1], outs[(False, "long")][1])
    for mode in (0, 1):
        a, b_ = outs[(True, mode)], outs[(False, mode)]
        assert np.array_equal(a[2], b_[2]), "log_duration"
        assert np.array_equal(a[1], b_[1])
        for x, y in zip(a[0], b_[0]):
            assert np.array_equal(x, y), mode


def test_fused_fp32_resblock_pair_is_bit_identical_to_the_two_kernel_path(pkg, full_bytes, monkeypatch):
    """rbpair32.hip (fp32, C = 32 / 64: conv1 -> LDS -> conv2 in one kernel) runs the MFMA chain and the epilogue expressions of two
    conv_mfma launches: the PCM must not move by a bit — ragged batch with very short members, windowed vocoder, both semantics modes,
    and utterances long enough that blocks of one launch start after others have finished (a fused block that read data another
    block had already overwritten would show up there)."""
    Ts = [30, 11, 40, 1, 2]
    ids = np.zeros((5, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 90 + b)
    long_ids = pkg.synth_ids(2, 700, ids_seed=4242)
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("VITS_NO_FUSE32", "1")
        with pkg.Model(full_bytes) as m:
            for mode in (0, 1):
                outs[(fused, mode, 0)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33)
                outs[(fused, mode, 1)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=33, vocoder_chunk_frames=24)
            outs[(fused, "long", 0)] = m.process_batch(long_ids, noise_seed=34, fixed_duration=2)
    for x, y in zip(outs[(True, "long", 0)][0], outs[(False, "long", 0)][0]):
        assert np.array_equal(x, y), "long utterance"
    for mode in (0, 1):
        for w in (0, 1):
            a, b_ = outs[(True, mode, w)], outs[(False, mode, w)]
            assert np.array_equal(a[1], b_[1])
            for x, y in zip(a[0], b_[0]):
                assert np.array_equal(x, y), (mode, w)
        for x, y in zip(outs[(True, mode, 0)][0], outs[(True, mode, 1)][0]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name,arith,_tol", ARITHS)
def test_converter_path_keeps_its_arithmetic_under_the_profiler_and_the_grouped_schedule(pkg, full_bytes, monkeypatch, name, arith, _tol):
    """on the fp32-layout (converter) path of a 16-bit mode (VITS_NO_GROUP16=1) the schedule must not change the arithmetic:
    the grouped launch and the fused fp32 pairs are fp32 kernels and stay out of a 16-bit run. Profiler on (which prefers the grouped
    schedule), VITS_RB_GROUP=1 and the plain three-stream run give the same PCM bit for bit — and it is NOT the fp32 PCM."""
    ids = pkg.synth_ids(3, 40, ids_seed=77)
    monkeypatch.setenv("VITS_NO_GROUP16", "1")
    outs = {}
    with pkg.Model(full_bytes) as m:
        f32 = m.process_batch(ids, noise_seed=12)
        m.set_arith(arith)
        outs["plain"] = m.process_batch(ids, noise_seed=12)
        m.prof_enable(True)
        outs["prof"] = m.process_batch(ids, noise_seed=12)
        m.prof_enable(False)
        names = [k["name"] for k in m.prof_report()["kernels"]]
        bad = [n for n in names if "resblock_group" in n or (n.startswith("hifigan_resblock_pair") and "|f" in n)]  # conv_group_kernel / rbpair32
        assert not bad, bad
    monkeypatch.setenv("VITS_RB_GROUP", "1")
    with pkg.Model(full_bytes) as m:
        m.set_arith(arith)
        outs["grouped"] = m.process_batch(ids, noise_seed=12)
    for k in ("prof", "grouped"):
        assert np.array_equal(outs["plain"][1], outs[k][1])
        for x, y in zip(outs["plain"][0], outs[k][0]):
            assert np.array_equal(x, y), k
    assert any(not np.array_equal(x, y) for x, y in zip(outs["plain"][0], f32[0]))


def test_whole_fp32_resblock_kernel_is_bit_identical_to_the_pair_path(pkg, full_bytes, monkeypatch):
    """rbblock32.hip (the 3-tap resblocks of the C = 32 / 64 stages as ONE fp32 kernel: the stream stays in registers across the three
    pairs) uses the same MFMA chain per output and the same epilogue expressions as three rbpair32 launches: the PCM must not move by a
    bit — ragged batch with one-token members, windowed vocoder, both semantics modes, 700-id utterances (many more blocks than the GPU
    holds at once), the serialised / grouped schedule of the profiler."""
    Ts = [30, 11, 40, 1, 2]
    ids = np.zeros((len(Ts), 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = _ids(T, 190 + b)
    long_ids = pkg.synth_ids(2, 700, ids_seed=4243)
    outs = {}
    for block in (True, False):
        if not block:
            monkeypatch.setenv("VITS_NO_RBBLOCK32", "1")
        with pkg.Model(full_bytes) as m:
            for mode in (0, 1):
                outs[(block, mode, 0)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=35)
                outs[(block, mode, 1)] = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=35, vocoder_chunk_frames=24)
            outs[(block, "long", 0)] = m.process_batch(long_ids, noise_seed=36, fixed_duration=2)
            m.prof_enable(True)
            outs[(block, "prof", 0)] = m.process_batch(ids, id_lengths=Ts, noise_seed=35)
            names = [k["name"] for k in m.prof_report()["kernels"]]
            assert any("hifigan_resblock_block|k3" in n for n in names) == block, names
            m.prof_enable(False)
    for key in [(m_, w) for m_ in (0, 1) for w in (0, 1)] + [("long", 0), ("prof", 0)]:
        a, b_ = outs[(True,) + key], outs[(False,) + key]
        assert np.array_equal(a[1], b_[1])
        for x, y in zip(a[0], b_[0]):
            assert np.array_equal(x, y), key
    for mode in (0, 1):
        for x, y in zip(outs[(True, mode, 0)][0], outs[(True, mode, 1)][0]):
            assert np.array_equal(x, y)


def test_kernel_tuning_knobs_belong_to_the_handle_that_read_them(pkg, full_bytes, monkeypatch):
    """The launch functions' tuning knobs (kernels.h KernelKnobs) are read when a model is loaded and travel with that handle: two handles
    loaded under different environments keep their own kernel choices while both are alive and their calls interleave — here the stride-8
    upsamplers of the f16 mode (VITS_NO_CONVT16L: conv16's polyphase epilogue instead of the four-phase streaming kernel) and the
    LayerNorm tile (VITS_LN_TW=64) — and the PCM is the same bit for bit."""
    ids = pkg.synth_ids(3, 48, ids_seed=77)
    monkeypatch.setenv("VITS_NO_CONVT16L", "1")
    monkeypatch.setenv("VITS_LN_TW", "64")
    a = pkg.Model(full_bytes)
    monkeypatch.delenv("VITS_NO_CONVT16L")
    monkeypatch.delenv("VITS_LN_TW")
    b = pkg.Model(full_bytes)
    try:
        outs, names = {}, {}
        for rnd in range(2):
            for tag, m in (("a", a), ("b", b)):
                if rnd == 0:
                    m.set_arith(pkg.ARITH_F16)
                    m.prof_enable(True)
                outs[(tag, rnd)] = m.process_batch(ids, noise_seed=5)
                names.setdefault(tag, set()).update(k["name"] for k in m.prof_report()["kernels"])
        assert not any("|SL" in n for n in names["a"]), sorted(names["a"])
        assert any("|SL" in n for n in names["b"]), sorted(names["b"])
        for rnd in range(2):
            assert np.array_equal(outs[("a", rnd)][1], outs[("b", rnd)][1])
            for x, y in zip(outs[("a", rnd)][0], outs[("b", rnd)][0]):
                assert np.array_equal(x, y)
    finally:
        a.close()
        b.close()


def test_16bit_conv_kernels_reproduce_torch_with_rounded_operands(pkg):
    """The HIP kernels of the 16-bit modes against a THIRD party (not the oracle): tests/golden/arith16_ops.npz = torch's Conv1d / ConvTranspose1d with both
    operands rounded to fp16 / bf16 and fp32 accumulation (make_golden.py `arith16_op_fixtures`; Q7: custom-ops.h:684-690). One conv differs from another
    correct implementation by fp32 summation order only — the fp32 bound of 2e-5 of RMS — and the fp32 kernels must miss the fixture by the rounding step."""
    from conftest import golden
    g = golden("arith16_ops.npz")
    for i in range(int(g["n_cases"][0])):
        meta = g["meta_%d" % i]
        kind, p1, crop, slope = int(meta[0]), int(meta[1]), int(meta[2]), float(meta[3]) / 1e6
        x, w, b = g["x_%d" % i], g["w_%d" % i], g["b_%d" % i]
        outs = {}
        for a in (pkg.ARITH_F16, pkg.ARITH_BF16, pkg.ARITH_F32):
            pkg.op_set_arith(a)
            if kind == 0:
                outs[a] = pkg.op_conv1d(x, w, b, dilation=p1, **(dict(pre_slope=slope) if slope != 1.0 else {}))
            else:
                outs[a] = pkg.op_conv_transpose1d(x, w, b, p1, crop, pre_slope=slope)
        for name, a in (("f16", pkg.ARITH_F16), ("bf16", pkg.ARITH_BF16)):
            y = g["y_%s_%d" % (name, i)]
            assert rel_err(outs[a], y) < 2e-5, (i, name, meta.tolist())
            assert rel_err(outs[pkg.ARITH_F32], y) > (2e-4 if name == "f16" else 1.5e-3), (i, name, meta.tolist())


@pytest.mark.parametrize("arch", ["tiny", "full"])
@pytest.mark.parametrize("name,arith,tol", ARITHS)
def test_full_path_16bit_reproduces_the_torch_operand_rounding_fixtures(pkg, arch, name, arith, tol):
    """Whole path in a 16-bit mode against transformers.VitsModel with torch hooks that round every conv operand of the flow and the vocoder
    (tests/golden/*_arith_*_taps.npz; see tests/test_oracle.py for what these fixtures can and cannot discriminate): stage-one taps at the fp32 bound,
    durations exact, downstream taps at the modes' rounding noise with the RMS deviation four times tighter; on the tiny architecture the flow
    output agrees far below the distance of the fp32 path."""
    from conftest import golden
    g = golden("%s_synth_arith_%s_taps.npz" % (arch, name))
    data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY if arch == "tiny" else pkg.SYNTH_FULL)
    kw = dict(mode=pkg.MODE_REFERENCE, noise_kind=pkg.NOISE_EXPLICIT, noise_dur=g["noise_dur"], noise_prior=g["noise_prior"], collect_taps=True)

    def dev(a, b):
        a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
        rms = np.sqrt((b ** 2).mean())
        return np.abs(a - b).max() / rms, np.sqrt(((a - b) ** 2).mean()) / rms

    with pkg.Model(data) as m:
        m.process_batch(g["ids"], **kw)
        z32 = m.tap("z_flow")
        m.set_arith(arith)
        pcm, lengths, frames = m.process_batch(g["ids"], **kw)
        np.testing.assert_array_equal(m.tap("durations"), g["durations"].ravel())
        assert lengths[0] == g["waveform"].size
        for tap in ("enc_out", "prior_mean", "log_duration", "z_p"):
            assert rel_err(m.tap(tap), g[tap]) < 1e-4, tap
        for tap in ("z_flow", "pre_tanh", "waveform"):
            mx, rms = dev(m.tap(tap), g[tap])
            assert mx < tol and rms < tol / 4, (tap, mx, rms)
        if arch == "tiny":
            assert dev(m.tap("z_flow"), g["z_flow"])[1] * 3 < dev(z32, g["z_flow"])[1]
