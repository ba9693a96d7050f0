"""Test helper (it runs the oracle, so it lives under tests/): CPU-oracle throughput vs thread count on this host (picks the cpu_baseline thread count)."""
import sys, time, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
spec = importlib.util.spec_from_file_location("vits_cpp_amd", os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
m = O.Model(pkg.synth_model_bytes(0x5EED, 0)); ids = pkg.synth_ids(1, 128)[0]
print("cpus", os.cpu_count())
for th in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 256]:
    t = time.time(); r = m.process_ids(ids, mode=0, noise_seed=4321, threads=th, taps=["waveform"]); dt = time.time() - t
    print(th, "threads %.2f s  %.0f samples/s" % (dt, r["waveform"].size / dt), flush=True)
