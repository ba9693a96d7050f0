"""Ad-hoc check at the longest supported input (2 x 2048 ids): fp32 and bf16 arithmetic give the same frame counts, the PCM is finite, and the windowed vocoder equals the whole-utterance run bit for bit. Run on a GPU box: python tools/long2048_check.py"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import importlib.util, os
spec = importlib.util.spec_from_file_location("vits_cpp_amd", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "vits.cpp_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
m.set_mode(pkg.MODE_REFERENCE)
ids = pkg.synth_ids(2, 2048, ids_seed=5)
for arith in (pkg.ARITH_F32, pkg.ARITH_BF16):
    m.set_arith(arith)
    pcm, lengths, frames = m.process_batch(ids, noise_kind=pkg.NOISE_COUNTER, noise_seed=7)
    pcw, lw, fw = m.process_batch(ids, noise_kind=pkg.NOISE_COUNTER, noise_seed=7, vocoder_chunk_frames=512)
    ok = all(np.array_equal(pcm[b][:lengths[b]], pcw[b][:lw[b]]) for b in range(2))
    print("arith", arith, "frames", list(frames), "finite", bool(np.isfinite(pcm[0][:lengths[0]]).all()), "rms", float(np.sqrt((pcm[0][:lengths[0]]**2).mean())), "windowed == whole", ok)
