"""Round-2 GPU tests through the C ABI: RCCL gather on hardware, BASELINE.json config 5 as written (two resident bf16-stored
models, 1024-id inputs, interleaved and concurrent calls), benchmark-batch utterances against the oracle, dispatcher
options (frames_only, noise_seed_offsets), the single-window streaming sink, and a plain-C caller."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def full_model(pkg, full_bytes):
    m = pkg.Model(full_bytes)
    yield m
    m.close()


# ---- multi-GPU: the path's only exchange, on hardware ---------------------------------------------------------------------
def _run_ranks(world, script, timeout=600):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [(p.returncode, o, e) for p, (o, e) in zip(procs, outs)]


def test_rccl_gather_of_pcm_matches_local_rows():
    """torch.distributed "nccl" (== RCCL) all-gather of the ragged PCM, fp32 and int16, contiguous and frames-balanced shards:
    every gathered row must equal the bits its owner computed. World = 2 when two devices are visible, otherwise one rank with
    the collective path forced (the launcher / init / gather code is the same)."""
    import torch
    world = 2 if torch.cuda.device_count() >= 2 else 1
    res = _run_ranks(world, os.path.join(ROOT, "tools", "dist_gather_check.py"))
    for r, (code, out, err) in enumerate(res):
        assert code == 0, f"rank {r}: {err[-3000:]}"
        assert f"rank {r} ok" in out


def test_bench_forced_dist_line(tmp_path):
    """bench.py with the RCCL exchange inside the timed step (VITS_BENCH_FORCE_DIST=1 on a 1-GPU box): n_gpus, value and
    the roofline block must be there, and the line must parse."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VITS_BENCH_FORCE_DIST"] = "1"
    env["VITS_BENCH_LAUNCH"] = "1"  # through the launcher (fresh child rank, JSON relayed), as `--gpus N` does for N > 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-extra-passes"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1, r.stdout[:400]  # ONE line, nothing else on stdout
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0 and line["scaling"] == "weak"
    assert line["ranks_seen"] == 1 and line["rccl_world_size"] == 1 and line["devices_seen"] == [0]  # counted through the communicator


# ---- dispatcher options ----------------------------------------------------------------------------------------------------
def test_frames_only_predicts_the_full_run(pkg, full_model):
    ids = pkg.synth_ids(6, 40)
    _, l_full, f_full = full_model.process_batch(ids, noise_seed=31)
    pcm, l_pre, f_pre = full_model.process_batch(ids, noise_seed=31, frames_only=True)
    assert pcm is None and np.array_equal(f_pre, f_full) and np.array_equal(l_pre, l_full)


def test_noise_seed_offsets_make_audio_independent_of_placement(pkg, full_model):
    """An utterance's audio depends on (ids, noise_seed + its offset) only — not on its row in the batch or on which other
    utterances share the call: what lets a dispatcher re-shard by frames without changing any output."""
    ids = pkg.synth_ids(5, 36)
    base, _, _ = full_model.process_batch(ids, noise_seed=900)
    perm = [3, 0, 4, 1, 2]
    moved, _, _ = full_model.process_batch(ids[perm], noise_seed=900, noise_seed_offsets=perm)
    for row, u in enumerate(perm):
        assert np.array_equal(moved[row], base[u])
    sub, _, _ = full_model.process_batch(ids[[4, 2]], noise_seed=900, noise_seed_offsets=[4, 2])
    assert rel_err(sub[0], base[4]) < 1e-5 and rel_err(sub[1], base[2]) < 1e-5


def test_streaming_sink_works_when_the_run_is_not_windowed(pkg, full_model):
    """a chunk size at least as long as the (unknowable) longest utterance used to FAIL the call after the front end
    had run. Now such a run streams as one window: one callback per utterance, offset 0, the whole utterance."""
    ids = pkg.synth_ids(3, 20)
    whole, lengths, frames = full_model.process_batch(ids, noise_seed=5)
    for kw in (dict(vocoder_chunk_frames=100000), dict()):
        got = {}

        def sink(utt, offset, pcm):
            assert offset == 0 and utt not in got
            got[utt] = pcm
            return False

        pcm, l2, _ = full_model.process_batch(ids, noise_seed=5, on_chunk=sink, **kw)
        assert sorted(got) == [0, 1, 2] and np.array_equal(l2, lengths)
        for b in range(3):
            assert np.array_equal(got[b], whole[b]) and np.array_equal(pcm[b], whole[b])
    with pytest.raises(pkg.VitsError, match="host copy"):
        full_model.process_batch(ids, on_chunk=lambda *a: False, skip_host_copy=True, out_device=1, out_device_stride=1 << 20)


# ---- BASELINE.json config 3: utterances of the actual benchmark batch against the oracle -------------------------------------
def test_benchmark_batch_utterances_match_the_oracle(pkg, oracle, full_model, full_bytes):
    """The bench's own batch (64 x 128 ids, ids seed 1234+u, noise seed 4321+u, reference mode, predicted durations): four of
    its utterances are run through the CPU oracle (~1 s each) — durations bit-exact (the ceil() boundaries of 128 predicted
    durations each), frame and sample counts exact, waveform within 1e-4 of RMS."""
    ids = pkg.synth_ids(64, 128)
    pcm, lengths, frames = full_model.process_batch(ids, noise_seed=4321, collect_taps=True)
    om = oracle.Model(full_bytes)
    for u in (0, 21, 42, 63):
        ref = om.process_ids(ids[u], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_COUNTER, noise_seed=4321 + u)
        np.testing.assert_array_equal(full_model.tap("durations", u), ref["durations"])
        assert frames[u] == int(ref["durations"].sum()) and lengths[u] == ref["waveform"].size == 256 * frames[u] + 294
        assert rel_err(full_model.tap("z_flow", u), ref["z_flow"]) < TOL
        assert rel_err(pcm[u], ref["waveform"]) < TOL


@pytest.mark.parametrize("arith", [0, 2], ids=["f32", "f16"])
def test_benchmark_batch_is_bit_reproducible_run_to_run(pkg, full_bytes, arith):
    """The bench batch (64 x 128 ids: several thousand blocks per launch, resblocks on concurrent streams, fused kernels whose
    blocks read their neighbours' halos) processed five times: every PCM sample identical. A block that reads data another
    block of the same launch overwrites shows up here as run-to-run noise (round 2 found one this way in the fused resblock pair)."""
    ids = pkg.synth_ids(64, 128)
    with pkg.Model(full_bytes) as m:
        m.set_arith(arith)
        first = None
        for rep in range(5):
            pcm, lengths, frames = m.process_batch(ids, noise_seed=4321)
            if first is None:
                first = ([x.copy() for x in pcm], lengths.copy())
                continue
            assert np.array_equal(lengths, first[1])
            for u, (x, y) in enumerate(zip(pcm, first[0])):
                assert np.array_equal(x, y), (rep, u)


def test_grouped_resblock_launches_are_bit_identical_to_separate_launches(pkg, full_bytes, monkeypatch):
    """conv_group_kernel (conv_mfma.hip): the same-position convolutions of the three resblocks of a C = 256 / C = 128 stage as ONE
    launch (11-tap blocks first, 3-tap blocks last; /root/reference/src/vits.cpp:622-635 runs the resblocks on the same input). Same
    kernel body per member, same order of the additions into the shared sum: the PCM of the benchmark batch and of a ragged,
    windowed batch must not move by a bit against VITS_NO_RB_GROUP=1 — with the profiler on (single stream) and off (fused
    resblocks on a side stream), both semantics modes."""
    ids = pkg.synth_ids(64, 128)
    Ts = [128, 3, 77, 128, 1, 40]
    outs = {}
    monkeypatch.setenv("VITS_RB_GROUP", "1")  # (grouped launches also without the profiler, where the library prefers its three streams)
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setenv("VITS_NO_RB_GROUP", "1")
        with pkg.Model(full_bytes) as m:
            outs[(grouped, "bench")] = m.process_batch(ids, noise_seed=4321)
            m.prof_enable(True)
            outs[(grouped, "bench_prof")] = m.process_batch(ids, noise_seed=4321)
            if grouped:
                names = [k["name"] for k in m.prof_report()["kernels"]]
                assert any("hifigan_resblock_group3" in n for n in names) and any("hifigan_resblock_group2" in n for n in names), names
            m.prof_enable(False)
            for mode in (0, 1):
                outs[(grouped, mode)] = m.process_batch(ids[:6], id_lengths=Ts, mode=mode, noise_seed=9)
                outs[(grouped, mode, "win")] = m.process_batch(ids[:6], id_lengths=Ts, mode=mode, noise_seed=9, vocoder_chunk_frames=100)
    for key in [k for k in outs if k[0]]:
        a, b_ = outs[key], outs[(False,) + key[1:]]
        assert np.array_equal(a[1], b_[1]), key
        for x, y in zip(a[0], b_[0]):
            assert np.array_equal(x, y), key
    for x, y in zip(outs[(True, "bench")][0], outs[(True, "bench_prof")][0]):
        assert np.array_equal(x, y)


# ---- BASELINE.json config 5 as written -----------------------------------------------------------------------------------
def test_config5_two_bf16_models_1024_ids_interleaved_and_concurrent(pkg, oracle):
    """Two resident models (seeds 0x5EED / 0xBEEF), conv weights stored as bf16 (type tag 2), 1024-id utterances:
    (a) each model against the oracle on a full 1024-id utterance (durations exact, waveform 1e-4);
    (b) interleaved calls do not disturb each other (bit-identical repeats);
    (c) windowed vocoder == whole utterance, bit for bit, on both;
    (d) the two handles driven from two threads at once give the same bits as serial calls (the boundary's promise that
        distinct handles may run concurrently)."""
    data = [pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL | pkg.SYNTH_BF16), pkg.synth_model_bytes(0xBEEF, pkg.SYNTH_FULL | pkg.SYNTH_BF16)]
    ids = [pkg.synth_ids(2, 1024, ids_seed=1234), pkg.synth_ids(2, 1024, ids_seed=91234)]
    models = [pkg.Model(d) for d in data]
    try:
        first = []
        for k in (0, 1, 0, 1):  # interleaved
            pcm, lengths, frames = models[k].process_batch(ids[k], noise_seed=50 + k)
            if len(first) < 2:
                first.append((pcm, lengths, frames))
            else:
                assert np.array_equal(lengths, first[k][1]) and all(np.array_equal(a, b) for a, b in zip(pcm, first[k][0]))
        for k in (0, 1):
            pcm, lengths, frames = first[k]
            assert (frames > 1024).all()
            ref = oracle.Model(data[k]).process_ids(ids[k][0], mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_COUNTER, noise_seed=50 + k)
            assert frames[0] == int(ref["durations"].sum()) and lengths[0] == ref["waveform"].size
            assert rel_err(pcm[0], ref["waveform"]) < TOL
            tiled, lt, _ = models[k].process_batch(ids[k], noise_seed=50 + k, vocoder_chunk_frames=256)
            assert np.array_equal(lt, lengths) and all(np.array_equal(a, b) for a, b in zip(tiled, pcm))
        results, errors = [None, None], []

        def worker(k):
            try:
                out = []
                for rep in range(3):
                    out.append(models[k].process_batch(ids[k], noise_seed=50 + k, vocoder_chunk_frames=512 if rep == 1 else 0))
                results[k] = out
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k in (0, 1):
            for pcm, lengths, _ in results[k]:
                assert np.array_equal(lengths, first[k][1]) and all(np.array_equal(a, b) for a, b in zip(pcm, first[k][0]))
    finally:
        for m in models:
            m.close()


# ---- a plain-C caller of the reference's five entry points --------------------------------------------------------------------
def test_c_caller_synthesises_through_the_reference_entry_points(pkg, oracle, tmp_path):
    """A C99 program (no ctypes) loads the file the REFERENCE'S exporter wrote (tests/golden/tiny_hf_export.ggml) with
    vits_model_load_from_file, calls vits_model_process(text) twice and prints sizes and checksums; the oracle, fed the same
    libstdc++ noise stream (seed 1, first draw of the process; /root/reference/src/vits.cpp:31), must give the same sample
    counts and the same audio."""
    path = os.path.join(ROOT, "tests", "golden", "tiny_hf_export.ggml")
    src = tmp_path / "synth.c"
    src.write_text(r'''
#include <stdio.h>
#include "vits.h"
int main(int argc, char** argv) {
    if (argc < 3) return 1;
    vits_model* m = vits_model_load_from_file(argv[1]);
    if (!m) { fprintf(stderr, "load: %s\n", vits_last_error()); return 2; }
    for (int rep = 0; rep < 2; ++rep) {
        vits_result r = vits_model_process(m, argv[2]);
        if (!r.data || r.size == 0) { fprintf(stderr, "process: %s\n", vits_last_error()); return 3; }
        double s = 0, a = 0;
        for (size_t i = 0; i < r.size; ++i) { s += r.data[i]; a += r.data[i] < 0 ? -r.data[i] : r.data[i]; }
        printf("%zu %.9e %.9e\n", r.size, s, a);
        vits_free_result(r);
    }
    vits_free_model(m);
    return 0;
}
''')
    exe = tmp_path / "synth"
    libdir = os.path.dirname(pkg.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                         "-L", libdir, "-lvits_hip", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    text = "hello there"
    run = subprocess.run([str(exe), path, text], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    rows = [ln.split() for ln in run.stdout.strip().splitlines()]
    with open(path, "rb") as fh:
        om = oracle.Model(fh.read())
    ids = om.tokenize(text)
    assert ids.size == 2 * len(text) + 1 - 2 * text.count(" ") or ids.size > 0
    oracle.lib().vo_reference_noise_seed(1)
    for rep in range(2):
        ref = om.process_ids(ids, mode=oracle.MODE_REFERENCE, noise_kind=oracle.NOISE_REFERENCE)["waveform"].astype(np.float64)
        assert int(rows[rep][0]) == ref.size
        assert abs(float(rows[rep][2]) - np.abs(ref).sum()) <= 2e-4 * np.abs(ref).sum()


@pytest.mark.parametrize("arith", [0, 2, 1], ids=["f32", "f16", "bf16"])
def test_fused_wavenet_layer_is_bit_identical_to_the_two_kernel_path(pkg, full_bytes, monkeypatch, arith):
    """wavenet32.hip (one flow WaveNet layer — gated conv, gate, 1x1 res/skip conv and both adds — as one kernel, fp32 and 16-bit
    operands) runs the MFMA chains, rounding points and epilogue expressions of the launches it replaces: z_flow and the PCM must not move by a bit. Ragged
    batch with very short members (one frame block and less), both semantics modes, and utterances long enough for several hundred
    32-frame blocks per launch (a block that read h columns another block had already overwritten would show up there)."""
    Ts = [30, 11, 40, 1, 2, 17]
    ids = np.zeros((6, 40), np.int32)
    rng = np.random.default_rng(5)
    for b, T in enumerate(Ts):
        ids[b, :T] = rng.integers(1, 38, size=T)
    long_ids = pkg.synth_ids(5, 900, ids_seed=99)  # 5 x 2700 frames = 5 x 85 blocks of 32 frames: the fused path (>= 384 blocks per launch)
    big = pkg.synth_ids(160, 40, ids_seed=7)       # 160 short ragged utterances (~3-4 blocks each): fused as well
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("VITS_NO_WN_FUSE", "1")
        with pkg.Model(full_bytes) as m:
            m.set_arith(arith)
            for mode in (0, 1):
                pcm, lengths, _ = m.process_batch(ids, id_lengths=Ts, mode=mode, noise_seed=21, collect_taps=True)
                outs[(fused, mode)] = (pcm, lengths, [m.tap("z_flow", u).copy() for u in range(len(Ts))])
            outs[(fused, "long")] = m.process_batch(long_ids, noise_seed=22, fixed_duration=3)
            outs[(fused, "big")] = m.process_batch(big, noise_seed=23)
    for key in ("long", "big"):
        assert np.array_equal(outs[(True, key)][1], outs[(False, key)][1])
        for x, y in zip(outs[(True, key)][0], outs[(False, key)][0]):
            assert np.array_equal(x, y), key
    for mode in (0, 1):
        a, b_ = outs[(True, mode)], outs[(False, mode)]
        assert np.array_equal(a[1], b_[1])
        for u in range(len(Ts)):
            assert np.array_equal(a[2][u], b_[2][u]), ("z_flow", mode, u)
            assert np.array_equal(a[0][u], b_[0][u]), ("pcm", mode, u)
