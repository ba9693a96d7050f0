"""Serial calls vs the single-handle pipeline (vits_model_submit_batch / vits_model_wait) on the benchmark batch: ms per batch.
usage: python tools/pipe_bench.py [--arith f16] [--steps 30] [--batch 64] [--ids 128]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--arith", default="f16")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ids", type=int, default=128)
ap.add_argument("--mode", choices=["both", "serial", "pipelined"], default="both", help="which schedule(s) to run (a profiler trace wants exactly one)")
ap.add_argument("--delay-ms", type=float, default=0.0, help="experiment: sleep this long before every submit of the pipelined loop (moves stage one later into the previous batch's vocoder)")
ap.add_argument("--stage-one", action="store_true", help="also time stage one alone (frames_only calls)")
a = ap.parse_args()
torch.zeros(1, device="cuda")  # (torch's HIP runtime first, as in bench.py: the other order leaves torch without a device)
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
m.set_arith({"f32": pkg.ARITH_F32, "f16": pkg.ARITH_F16, "bf16": pkg.ARITH_BF16}[a.arith])
ids = pkg.synth_ids(a.batch, a.ids)
cap = 256 * 8 * a.ids + 294
bufs = [torch.empty((a.batch, cap), dtype=torch.float32, device="cuda") for _ in range(2)]
kw = dict(noise_seed=4321, out_device_stride=cap, skip_host_copy=True)
for k in range(3):
    m.process_batch(ids, out_device=bufs[k % 2].data_ptr(), keep_pcm=False, **kw)
res = {}
for rep in range(2):
    if a.mode != "pipelined":
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(a.steps):
            _, lengths, _ = m.process_batch(ids, out_device=bufs[k % 2].data_ptr(), keep_pcm=False, **kw)
        torch.cuda.synchronize(); res["serial"] = (time.perf_counter() - t) / a.steps * 1e3
    if a.mode == "serial":
        res["pipelined"] = float("nan")
        continue
    res.setdefault("serial", float("nan"))
    torch.cuda.synchronize(); t = time.perf_counter()
    m.submit_batch(ids, out_device=bufs[0].data_ptr(), **kw)
    t_sub = t_wait = 0.0
    for k in range(1, a.steps):
        if a.delay_ms > 0:
            time.sleep(a.delay_ms * 1e-3)
        ta = time.perf_counter()
        m.submit_batch(ids, out_device=bufs[k % 2].data_ptr(), **kw)
        tb = time.perf_counter()
        _, lengths, _ = m.wait(keep_pcm=False)
        t_sub += tb - ta
        t_wait += time.perf_counter() - tb
    res["submit_ms"] = t_sub / max(a.steps - 1, 1) * 1e3  # host time inside submit (stage one + the frame-count read + queueing stage two)
    res["wait_ms"] = t_wait / max(a.steps - 1, 1) * 1e3   # host time blocked in wait: the slack before the previous batch's vocoder ends
    _, lengths, _ = m.wait(keep_pcm=False)
    torch.cuda.synchronize(); res["pipelined"] = (time.perf_counter() - t) / a.steps * 1e3
if a.stage_one:
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(a.steps):
            m.process_batch(ids, noise_seed=4321, frames_only=True, keep_pcm=False)
        torch.cuda.synchronize(); res["stage1"] = (time.perf_counter() - t) / a.steps * 1e3
samples = int(lengths.sum())
print(f"{a.arith} batch {a.batch} x {a.ids}: serial {res['serial']:.3f} ms ({samples / res['serial'] / 1e3:.1f} M/s)  pipelined {res['pipelined']:.3f} ms "
      f"({samples / res['pipelined'] / 1e3:.1f} M/s)" + (f"  stage one alone {res['stage1']:.3f} ms" if "stage1" in res else "") + (f"  [submit {res['submit_ms']:.2f} ms, wait {res['wait_ms']:.2f} ms]" if "submit_ms" in res else "") + (f"  delay {a.delay_ms} ms" if a.delay_ms else "") + "  env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VITS_")))
m.close()
