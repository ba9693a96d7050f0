"""Model load time (vits_model_load_from_bytes: parse, shape checks, MFMA-fragment packing, upload) and the first call. usage: python tools/load_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
pkg.set_device(0)
for rep in range(3):
    t = time.perf_counter(); m = pkg.Model(data); t1 = time.perf_counter() - t
    ids = pkg.synth_ids(1, 128)[0]
    t = time.perf_counter(); m.process_ids(ids); t2 = time.perf_counter() - t
    t = time.perf_counter(); m.process_ids(ids); t3 = time.perf_counter() - t
    t = time.perf_counter(); m.set_arith(pkg.ARITH_F16); t4 = time.perf_counter() - t
    print(f"load {t1*1e3:.0f} ms ({len(data)/1e6:.1f} MB file) | first call {t2*1e3:.1f} ms | second {t3*1e3:.2f} ms | set_arith(f16) {t4*1e3:.0f} ms")
    m.close()
