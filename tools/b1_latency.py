"""Batch-1 latency breakdown: wall per call (median of 20) with 1 vs 3 resblock streams, and the HIP-event kernel list."""
import sys, os, time, json, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
ids = pkg.synth_ids(1, T)
for _ in range(3):
    m.process_batch(ids, noise_seed=1)
t = []
for _ in range(20):
    t0 = time.perf_counter(); m.process_batch(ids, noise_seed=1); t.append(time.perf_counter() - t0)
print("streams", os.environ.get("VITS_RB_STREAMS", "default"), "wall ms median", 1e3 * np.median(t), "min", 1e3 * min(t))
m.prof_enable(True); m.prof_reset()
t0 = time.perf_counter(); m.process_batch(ids, noise_seed=1); dt = time.perf_counter() - t0
rep = m.prof_report()["kernels"]
groups = {}
for k in rep:
    g = k["name"].split("|")[0]
    a = groups.setdefault(g, [0, 0.0]); a[0] += k["calls"]; a[1] += k["ms"]
print("profiled (serial) wall ms", 1e3 * dt, "kernel ms", sum(k["ms"] for k in rep), "launches", sum(k["calls"] for k in rep))
for g, (c, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} ms {c:5d} calls {1e3*ms/c:8.1f} us/call  {g}")
