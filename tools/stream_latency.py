"""Time to first audio with the windowed vocoder (vits_process_opts.vocoder_chunk_frames + on_chunk), batch 1.
usage: python tools/stream_latency.py [ids ...]   -> one JSON line per (ids, chunk)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package

pkg = load_package()
model = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
for T in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    ids = pkg.synth_ids(1, T)
    for _ in range(3):
        model.process_batch(ids, noise_seed=1)
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); pcm, lengths, frames = model.process_batch(ids, noise_seed=1); t.append(time.perf_counter() - t0)
    whole_ms = 1e3 * float(np.median(t))
    audio_s = float(lengths[0]) / model.sampling_rate
    print(json.dumps({"ids": T, "frames": int(frames[0]), "audio_s": audio_s, "chunk_frames": 0, "total_ms": whole_ms, "first_audio_ms": whole_ms}))
    for chunk in (16, 32, 64, 128, 256):
        if chunk >= frames[0]:
            continue
        firsts, totals = [], []
        for _ in range(10):
            stamps = []
            t0 = time.perf_counter()
            model.process_batch(ids, noise_seed=1, vocoder_chunk_frames=chunk, on_chunk=lambda u, off, p: stamps.append(time.perf_counter()) and False)
            totals.append(time.perf_counter() - t0)
            firsts.append(stamps[0] - t0)
        print(json.dumps({"ids": T, "frames": int(frames[0]), "audio_s": audio_s, "chunk_frames": chunk, "chunk_audio_ms": 1e3 * chunk * 256 / model.sampling_rate,
                          "total_ms": 1e3 * float(np.median(totals)), "first_audio_ms": 1e3 * float(np.median(firsts)), "windows": len(stamps)}))
