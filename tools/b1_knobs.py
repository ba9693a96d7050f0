"""Batch-1 (config 2) A/B of environment knobs in ONE process: the knobs are read when a model is loaded (Engine::knobs), so every configuration is a
fresh handle. usage: python tools/b1_knobs.py [f16|f32|both] "KNOB=V KNOB2=V2" "KNOB3=V" ...   ("" = the default configuration is always measured first and last)
Prints the median / minimum wall time of 60 counter-noise process_batch calls (one 128-id utterance) per configuration and arithmetic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
arg = sys.argv[1] if len(sys.argv) > 1 else "both"
ariths = {"f16": [pkg.ARITH_F16], "f32": [pkg.ARITH_F32], "both": [pkg.ARITH_F16, pkg.ARITH_F32]}[arg]
configs = [""] + list(sys.argv[2:]) + [""]
data = pkg.synth_model_bytes(0x5EED, 0)
ids = pkg.synth_ids(1, 128)
ref = {}
for cfg in configs:
    kv = dict(x.split("=", 1) for x in cfg.split()) if cfg else {}
    for k, v in kv.items():
        os.environ[k] = v
    m = pkg.Model(data)
    for k in kv:
        del os.environ[k]
    for a in ariths:
        m.set_arith(a)
        for _ in range(5):
            r = m.process_batch(ids, noise_seed=1)
        t = []
        for _ in range(60):
            t0 = time.perf_counter(); r = m.process_batch(ids, noise_seed=1); t.append(time.perf_counter() - t0)
        pcm = np.asarray(r[0][0] if isinstance(r, tuple) else r[0])
        same = ""
        if a in ref:
            same = "bit-identical to default" if pcm.shape == ref[a].shape and np.array_equal(pcm, ref[a]) else "DIFFERS from default"
        else:
            ref[a] = pcm.copy()
        print("%-44s %s  median %.3f ms  min %.3f ms  %s" % (cfg or "(default)", "f16" if a == pkg.ARITH_F16 else "f32", 1e3 * np.median(t), 1e3 * min(t), same), flush=True)
    m.close()
