"""VITS_ARITH_F32_SPLIT (round 6; csrc/conv_split.hip): fp32-accurate ResBlock convolutions of the vocoder's wide stages on the bf16 matrix cores by operand
splitting — weights (fp16 values) as two bf16 pieces, fp32 activations as three, the five significant cross products accumulated in fp32
(/root/reference/src/include/custom-ops.h:680-694 is the conv it computes; /root/reference/src/vits.cpp:545-581 the resblock). Not the fmaf chain's bits: its parity bar
is the fp32 bar itself — every float tap against the fp32 ORACLE within the tolerance the exact-fp32 GPU path is held to (1e-4 of RMS), durations / frames / samples
exact — plus: the distance to the exact-fp32 GPU path is at rounding-noise level, and the mode keeps the invariances that do not depend on the summation order."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_split_arithmetic_matches_the_fp32_oracle_like_the_fp32_path(pkg, oracle, full_bytes):
    om = oracle.Model(full_bytes)
    Ts = [40, 17, 33]
    ids = np.zeros((3, 40), np.int32)
    for b, T in enumerate(Ts):
        ids[b, :T] = pkg.synth_ids(1, T, ids_seed=310 + b)[0]
    with pkg.Model(full_bytes) as m:
        for mode, omode in ((pkg.MODE_REFERENCE, oracle.MODE_REFERENCE), (pkg.MODE_HF, oracle.MODE_HF)):
            m.set_arith(pkg.ARITH_F32)
            exact, le, fe = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=17)
            m.set_arith(pkg.ARITH_F32_SPLIT)
            assert m.arith == pkg.ARITH_F32_SPLIT
            pcm, lengths, frames = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=17, collect_taps=True)
            assert np.array_equal(lengths, le) and np.array_equal(frames, fe)
            worst_gpu = worst_or = 0.0
            for b, T in enumerate(Ts):
                ref = om.process_ids(ids[b, :T], mode=omode, noise_kind=oracle.NOISE_COUNTER, noise_seed=17 + b)
                assert np.array_equal(m.tap("durations", b), ref["durations"])
                assert pcm[b].size == ref["waveform"].size
                for name in ("z_flow", "pre_tanh", "waveform"):
                    e = rel_err(m.tap(name, b), ref[name])
                    worst_or = max(worst_or, e)
                    assert e < 1e-4, (mode, b, name, e)
                worst_gpu = max(worst_gpu, rel_err(pcm[b], exact[b]))
                assert not np.array_equal(pcm[b], exact[b])  # (the split kernels really ran: another summation order)
            assert worst_gpu < 5e-5, worst_gpu  # split vs the exact-fp32 GPU path: rounding noise of an fp32 accumulation (measured ~1e-5)
            print("split arithmetic, mode %d: max tap error vs oracle %.2e, PCM vs the exact-fp32 path %.2e" % (mode, worst_or, worst_gpu))


def test_split_arithmetic_keeps_batch_and_window_invariance(pkg, full_bytes):
    """A sample's products and their order do not depend on the batch it sits in or on the vocoder window that computes it — in this arithmetic as well."""
    ids = pkg.synth_ids(4, 64, ids_seed=91)
    lens = np.array([64, 9, 50, 31], np.int32)
    with pkg.Model(full_bytes) as m:
        m.set_arith(pkg.ARITH_F32_SPLIT)
        whole, lw, _ = m.process_batch(ids, id_lengths=lens, noise_seed=4)
        for b in range(4):
            alone, la, _ = m.process_batch(ids[b:b + 1, :lens[b]], noise_seed=4, noise_seed_offsets=np.array([b], np.int32))
            assert la[0] == lw[b] and np.array_equal(alone[0], whole[b]), b
        tiled, lt, _ = m.process_batch(ids, id_lengths=lens, noise_seed=4, vocoder_chunk_frames=37)
        assert np.array_equal(lt, lw)
        for b in range(4):
            assert np.array_equal(tiled[b], whole[b]), b
        # and back: the exact mode is untouched by having been in the split mode
        m.set_arith(pkg.ARITH_F32)
        again, _, _ = m.process_batch(ids, id_lengths=lens, noise_seed=4)
    with pkg.Model(full_bytes) as m2:
        fresh, _, _ = m2.process_batch(ids, id_lengths=lens, noise_seed=4)
    for b in range(4):
        assert np.array_equal(again[b], fresh[b])


def test_split_arithmetic_on_the_benchmark_utterance(pkg, full_bytes):
    """utterance 0 of the benchmark batch against the patched-transformers reference-mode fixture (tests/golden/bench_utt0_refmode_taps.npz): durations and sample
    count exact, the float taps within the bound the exact-fp32 path is held to."""
    from conftest import golden
    g = golden("bench_utt0_refmode_taps.npz")
    ids = pkg.synth_ids(1, 128, ids_seed=int(g["ids_seed"][0]))
    np.testing.assert_array_equal(ids[0], g["ids"])
    d = int(g["decimate"][0])
    with pkg.Model(full_bytes) as m:
        m.set_arith(pkg.ARITH_F32_SPLIT)
        pcm, lengths, frames = m.process_batch(ids, mode=pkg.MODE_REFERENCE, noise_seed=int(g["noise_seed"][0]), collect_taps=True)
        np.testing.assert_array_equal(m.tap("durations", 0), g["durations"].ravel())
        assert lengths[0] == int(g["waveform_len"][0]) and frames[0] == int(g["durations"].sum())
        assert rel_err(m.tap("z_flow", 0), g["z_flow"]) < 2e-4
        assert rel_err(m.tap("pre_tanh", 0)[::d], g["pre_tanh_decimated"]) < 2e-4
        assert rel_err(pcm[0][::d], g["waveform_decimated"]) < 2e-4
