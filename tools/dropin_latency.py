"""Latency of the reference's own entry point (vits_model_process_ids with the reference noise stream: libstdc++ engine on the host) against the
counter-noise batch-1 call, fp32 and f16. usage: python tools/dropin_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
ids = pkg.synth_ids(1, 128)[0]
for arith, name in ((pkg.ARITH_F32, "f32"), (pkg.ARITH_F16, "f16")):
    m.set_arith(arith)
    for _ in range(5):
        m.process_ids(ids)
        m.process_batch(ids, noise_seed=1)
    n = 40
    t = time.perf_counter()
    for _ in range(n):
        pcm = m.process_ids(ids)
    a = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n):
        m.process_batch(ids, noise_seed=1)
    b = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n):
        m.process_batch(ids, noise_seed=1, keep_pcm=False, skip_host_copy=True)
    c = (time.perf_counter() - t) / n * 1e3
    print(f"{name}: vits_model_process_ids (reference noise, host PCM) {a:.3f} ms | process_batch counter noise, host PCM {b:.3f} ms | counter noise, PCM left on the device {c:.3f} ms | samples {pcm.size}")
m.close()
