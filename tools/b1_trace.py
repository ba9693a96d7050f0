"""Batch-1 run for rocprofv3 --kernel-trace: 20 calls of one 128-id utterance (after warmup)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
ids = pkg.synth_ids(1, T)
for _ in range(3):
    m.process_batch(ids, noise_seed=1)
t0 = time.perf_counter()
for _ in range(20):
    m.process_batch(ids, noise_seed=1)
print("wall ms per call", (time.perf_counter() - t0) / 20 * 1e3)
