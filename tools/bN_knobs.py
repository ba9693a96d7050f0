"""A/B of environment knobs at several small batch sizes in ONE process (f16 arithmetic): usage python tools/bN_knobs.py "1 2 4 8" "KNOB=V ..." ... ; see tools/b1_knobs.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
batches = [int(x) for x in sys.argv[1].split()]
arith = pkg.ARITH_F32 if os.environ.get("BN_ARITH") == "f32" else pkg.ARITH_F16
configs = [""] + list(sys.argv[2:]) + [""]
data = pkg.synth_model_bytes(0x5EED, 0)
for cfg in configs:
    kv = dict(x.split("=", 1) for x in cfg.split()) if cfg else {}
    os.environ.update(kv)
    m = pkg.Model(data)
    for k in kv:
        del os.environ[k]
    m.set_arith(arith)
    row = []
    for B in batches:
        ids = pkg.synth_ids(B, 128)
        for _ in range(4):
            m.process_batch(ids, noise_seed=1, keep_pcm=False)
        t = []
        for _ in range(40):
            t0 = time.perf_counter(); m.process_batch(ids, noise_seed=1, keep_pcm=False); t.append(time.perf_counter() - t0)
        row.append("b%d %.3f" % (B, 1e3 * np.median(t)))
    print("%-60s %s" % (cfg or "(default)", "  ".join(row)), flush=True)
    m.close()
