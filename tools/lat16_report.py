"""Developer helper: per-launch-shape times of the small-grid convs at batch 1 (the engine's per-kernel profiler)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
ids = pkg.synth_ids(1, 128)
for _ in range(3):
    m.process_batch(ids, noise_seed=1)
m.prof_enable(True)
N = 10
for _ in range(N):
    m.process_batch(ids, noise_seed=1)
rep = m.prof_report()
tot = 0.0
for k in sorted(rep["kernels"], key=lambda k: -k["ms"]):
    if "|t6|" in k["name"] or "|t5|" in k["name"]:
        print("%-70s calls/step %5.1f  us per call %7.1f" % (k["name"][:70], k["calls"] / N, 1e3 * k["ms"] / k["calls"]))
    tot += k["ms"]
print("sum of kernel time per step %.3f ms" % (tot / N))
m.close()
