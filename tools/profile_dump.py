"""Developer helper: runs the benchmark workload once with the HIP-event profiler and prints every kernel entry."""
import sys, json, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vits_cpp_amd", os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = 3
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
if len(sys.argv) > 3:
    m.set_arith({"f32": 0, "bf16": 1, "f16": 2}[sys.argv[3]])
ids = pkg.synth_ids(B, T)
for _ in range(2):
    m.process_batch(ids, noise_seed=4321, skip_host_copy=True)
m.prof_enable(True); m.prof_reset()
import time
t0 = time.perf_counter()
for _ in range(steps):
    _, lengths, frames = m.process_batch(ids, noise_seed=4321, skip_host_copy=True)
dt = (time.perf_counter() - t0) / steps
rep = m.prof_report()["kernels"]
tot = sum(k["ms"] for k in rep) / steps
print(f"step {dt*1e3:.2f} ms, kernels {tot:.2f} ms, samples/s {lengths.sum()/dt:.3e}, frames mean {frames.mean():.1f}")
for k in sorted(rep, key=lambda k: -k["ms"]):
    ms = k["ms"] / steps
    tf = k["flop"] / (k["ms"] * 1e-3) / 1e12 if k["flop"] else 0
    gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["bytes"] else 0
    print(f"{ms:8.3f} ms  {k['calls']//steps:4d} calls  {tf:7.1f} TF  {gbs:7.0f} GB/s(alg)  {k['name']}")
