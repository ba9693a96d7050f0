"""Prints a hash of the PCM of a fixed ragged batch; run under different VITS_* knob settings: the hashes must be equal."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
arith = {"f32": pkg.ARITH_F32, "f16": pkg.ARITH_F16, "bf16": pkg.ARITH_BF16}[os.environ.get("VITS_KNOB_ARITH", "f32")]
if arith != pkg.ARITH_F32:
    m.set_arith(arith)
ids = pkg.synth_ids(6, 48)
lens = np.array([48, 7, 33, 48, 1, 20], np.int32)
h = hashlib.sha256()
for mode in (0, 1):
    pcm, lengths, frames = m.process_batch(ids, id_lengths=lens, mode=mode, noise_seed=5)
    for p in pcm:
        h.update(p.tobytes())
if arith != pkg.ARITH_F32:  # longer utterances as well: more than one tile per utterance on every vocoder stage, windowed and whole
    ids2 = pkg.synth_ids(3, 300, ids_seed=77)
    for chunk in (0, 150):
        pcm, _, _ = m.process_batch(ids2, id_lengths=np.array([300, 41, 160], np.int32), noise_seed=6, vocoder_chunk_frames=chunk)
        for p in pcm:
            h.update(p.tobytes())
print(h.hexdigest()[:16])
