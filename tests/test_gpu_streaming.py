"""Windowed vocoder / chunked emission through the C ABI: vits_process_opts.vocoder_chunk_frames and
.on_chunk. The contract is BIT-IDENTITY with the whole-utterance run (which the other GPU tests pin against the oracle):
a window computes its owned frames plus the vocoder's exact receptive field on both sides, and every emitted sample
accumulates the same products in the same order."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_model(pkg, full_bytes):
    m = pkg.Model(full_bytes)
    yield m
    m.close()


def ragged_ids(pkg, lens, seed=0):
    stride = max(lens)
    ids = np.zeros((len(lens), stride), np.int32)
    for b, n in enumerate(lens):
        ids[b, :n] = pkg.synth_ids(1, n, ids_seed=1234 + 17 * seed + b)[0]
    return ids, np.asarray(lens, np.int32)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("chunk", [16, 37, 64])
def test_windowed_vocoder_is_bit_identical(pkg, full_model, mode, chunk):
    ids, lens = ragged_ids(pkg, [40, 5, 64, 17, 1])
    kw = dict(id_lengths=lens, mode=mode, noise_seed=55)
    whole, lw, fw = full_model.process_batch(ids, **kw)
    tiled, lt, ft = full_model.process_batch(ids, vocoder_chunk_frames=chunk, **kw)
    assert np.array_equal(lw, lt) and np.array_equal(fw, ft)
    assert int(fw.max()) > chunk  # more than one window, or the test says nothing
    for a, b in zip(whole, tiled):
        assert np.array_equal(a, b)


def test_windowed_vocoder_pinned_and_device_output(pkg, full_model):
    """Pinned durations (no host read of the frame counts) with a chunk that divides the length exactly, and one that
    leaves a 1-frame last window."""
    ids = pkg.synth_ids(3, 33)
    whole, lw, _ = full_model.process_batch(ids, fixed_duration=2, noise_seed=8)
    for chunk in (33, 22, 65):
        tiled, lt, _ = full_model.process_batch(ids, fixed_duration=2, noise_seed=8, vocoder_chunk_frames=chunk)
        assert np.array_equal(lw, lt)
        for a, b in zip(whole, tiled):
            assert np.array_equal(a, b)


def test_chunks_tile_every_utterance_in_order(pkg, full_model):
    ids, lens = ragged_ids(pkg, [48, 9, 30], seed=3)
    got = {b: [] for b in range(3)}
    stamps = []

    def sink(utt, offset, pcm):
        got[utt].append((offset, pcm))
        stamps.append(time.perf_counter())
        return False

    t0 = time.perf_counter()
    pcm, lengths, frames = full_model.process_batch(ids, id_lengths=lens, noise_seed=21, vocoder_chunk_frames=24, on_chunk=sink)
    t1 = time.perf_counter()
    whole, _, _ = full_model.process_batch(ids, id_lengths=lens, noise_seed=21)
    for b in range(3):
        pos = 0
        for offset, chunk in got[b]:
            assert offset == pos  # in order, no gaps, no overlaps
            pos += len(chunk)
        assert pos == lengths[b]
        assert np.array_equal(np.concatenate([c for _, c in got[b]]), whole[b])
        assert np.array_equal(pcm[b], whole[b])
        # every chunk but the last is 24 frames = 24 * 256 samples
        assert all(len(c) == 24 * 256 for _, c in got[b][:-1])
    assert len(got[0]) == -(-int(frames[0]) // 24)
    assert stamps[0] < t1 and stamps[0] > t0


def test_callback_abort_and_misuse(pkg, full_model):
    ids = pkg.synth_ids(1, 40)
    calls = []

    def stop(utt, offset, pcm):
        calls.append(offset)
        return True

    with pytest.raises(pkg.VitsError, match="aborted"):
        full_model.process_batch(ids, vocoder_chunk_frames=16, on_chunk=stop)
    assert calls == [0]

    def boom(utt, offset, pcm):
        raise KeyError("sink failed")

    with pytest.raises(KeyError):
        full_model.process_batch(ids, vocoder_chunk_frames=16, on_chunk=boom)
    calls.clear()
    with pytest.raises(pkg.VitsError, match="aborted"):
        full_model.process_batch(ids, on_chunk=stop)  # no chunking requested: one window, the sink still gets (and may abort) it
    assert calls == [0]
    # the model is still usable afterwards
    pcm, _, _ = full_model.process_batch(ids, vocoder_chunk_frames=16)
    assert np.isfinite(pcm[0]).all()


def test_long_form_1024_ids_windowed(pkg, full_model):
    """Config 5 input length through 256-frame windows: same PCM, activations bounded by the window."""
    ids = pkg.synth_ids(2, 1024)
    whole, lw, fw = full_model.process_batch(ids, noise_seed=9)
    tiled, lt, _ = full_model.process_batch(ids, noise_seed=9, vocoder_chunk_frames=256)
    assert np.array_equal(lw, lt) and int(fw.max()) > 1024
    for a, b in zip(whole, tiled):
        assert np.array_equal(a, b)
