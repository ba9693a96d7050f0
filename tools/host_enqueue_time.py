"""Batch 1: how long the host needs to ENQUEUE a call (async + out_device: the call returns when everything is queued) against the call's total time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
import torch
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, 0))
for name, ar in (("f16", pkg.ARITH_F16), ("f32", pkg.ARITH_F32)):
    m.set_arith(ar)
    ids = pkg.synth_ids(1, 128)
    buf = torch.empty(1 * 200000, dtype=torch.float32, device="cuda")
    for _ in range(5):
        m.process_batch(ids, noise_seed=1, out_device=buf.data_ptr(), out_device_stride=200000, skip_host_copy=True)
    te, tt = [], []
    for _ in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.process_batch(ids, noise_seed=1, out_device=buf.data_ptr(), out_device_stride=200000, skip_host_copy=True, async_=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        te.append(t1 - t0); tt.append(t2 - t0)
    print("%s: host returns after %.3f ms (median), device done after %.3f ms" % (name, 1e3 * np.median(te), 1e3 * np.median(tt)))
