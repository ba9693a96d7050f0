"""Soak / leak check: thousands of calls through every entry-point family, host RSS and free device memory before and after.
usage: python tools/soak.py [--calls 4000]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=4000); a = ap.parse_args()
pkg = load_package()
hip = C.CDLL("libamdhip64.so")
def rss_mb():
    with open("/proc/self/status") as f:
        for ln in f:
            if ln.startswith("VmRSS"): return int(ln.split()[1]) / 1024.0
def dev_free_mb():
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot)); return fr.value / 2**20
m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
ids1 = pkg.synth_ids(1, 128)[0]; ids8 = pkg.synth_ids(8, 96)
def round_(n):
    for i in range(n):
        m.process_ids(ids1[: 16 + (i * 7) % 112])                      # the reference's own call, varying lengths (helper thread, rewinds, growth)
    for i in range(n // 20):
        m.process_batch(ids8, noise_seed=i)                             # batches, host PCM
        m.submit_batch(ids8, noise_seed=i); m.submit_batch(ids8, noise_seed=i + 1); m.wait(); m.wait()
        m.process_batch(ids8[:2], noise_kind=pkg.NOISE_REFERENCE)       # B > 1 with the reference stream (old path)
    for ar in (pkg.ARITH_F16, pkg.ARITH_F32):
        m.set_arith(ar)
        for i in range(n // 40): m.process_batch(ids8, noise_seed=i, vocoder_chunk_frames=64)
    m.set_ggml_tables(1); m.process_batch(ids8[:2], frames_only=True); m.set_ggml_tables(0)
round_(200)  # warm: arenas, pools, fragments
r0, d0 = rss_mb(), dev_free_mb()
t = time.time(); round_(a.calls); dt = time.time() - t
r1, d1 = rss_mb(), dev_free_mb()
print(f"soak: {a.calls} reference-API calls + batches / pipeline / windows / arithmetic switches in {dt:.1f} s; host RSS {r0:.0f} -> {r1:.0f} MB, free device memory {d0:.0f} -> {d1:.0f} MB")
assert r1 - r0 < 64 and d0 - d1 < 64, "memory grew"
m.close()
print("soak ok")
