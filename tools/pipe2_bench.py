"""Experiment: TWO handles, each running the pipelined loop on its own host thread (stage-two overlap on top of the stage-one overlap)."""
import argparse, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--arith", default="f16")
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
torch.zeros(1, device="cuda")
pkg = load_package()
data = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
ms = [pkg.Model(data) for _ in range(2)]
ar = {"f32": pkg.ARITH_F32, "f16": pkg.ARITH_F16, "bf16": pkg.ARITH_BF16}[a.arith]
for m in ms:
    m.set_arith(ar)
ids = pkg.synth_ids(64, 128)
cap = 256 * 8 * 128 + 294
bufs = [[torch.empty((64, cap), dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(2)]
kw = dict(noise_seed=4321, out_device_stride=cap, skip_host_copy=True)
def loop(i, n):
    m = ms[i]
    m.submit_batch(ids, out_device=bufs[i][0].data_ptr(), **kw)
    for k in range(1, n):
        m.submit_batch(ids, out_device=bufs[i][k % 2].data_ptr(), **kw)
        m.wait(keep_pcm=False)
    m.wait(keep_pcm=False)
for i in range(2):
    loop(i, 3)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    loop(0, a.steps)
    torch.cuda.synchronize(); one = (time.perf_counter() - t) / a.steps * 1e3
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=loop, args=(i, a.steps)) for i in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); two = (time.perf_counter() - t) / (2 * a.steps) * 1e3
print(f"{a.arith}: one pipelined handle {one:.3f} ms per batch; two pipelined handles {two:.3f} ms per batch")
for m in ms: m.close()
