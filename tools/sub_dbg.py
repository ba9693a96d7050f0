import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
if os.environ.get("DBG_ORDER") == "bench":
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    pkg = bench.load_package()
    pkg.set_device(0)
else:
    torch.zeros(1, device="cuda")
    pkg = bench.load_package()
mb = pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)
m = pkg.Model(mb)
m.set_mode(pkg.MODE_REFERENCE)
stage = sys.argv[1] if len(sys.argv) > 1 else "fresh"
ids = pkg.synth_ids(64, 128)
cap = 256 * 8 * 128 + 294
buf = torch.empty((64, cap), dtype=torch.float32, device="cuda")
kw = dict(noise_seed=4321, out_device=buf.data_ptr(), out_device_stride=cap, skip_host_copy=True, keep_pcm=False)
if stage in ("prof", "all"):
    m.prof_enable(True)
    for _ in range(int(os.environ.get("DBG_PROF_STEPS", "3"))):
        m.process_batch(ids, **kw)
    m.prof_enable(False)
    m.prof_report()
if stage in ("pinned", "all"):
    m.process_batch(ids, fixed_duration=2, **kw)
    m.process_batch(ids, **kw)
if stage in ("host", "all"):
    kw2 = dict(kw); kw2["skip_host_copy"] = False
    for _ in range(3):
        m.process_batch(ids, **kw2)
if stage in ("m2", "all"):
    os.environ["VITS_NO_RB_GROUP"] = "1"
    m2 = pkg.Model(mb)
    del os.environ["VITS_NO_RB_GROUP"]
    m2.process_batch(ids, **kw)
    m2.prof_reset(); m2.prof_enable(True); m2.process_batch(ids, **kw); m2.prof_enable(False); m2.prof_report()
    m2.close()
r = bench.sub_results(pkg, torch, m, mb, pkg.MODE_REFERENCE)
for k, v in r.items():
    if isinstance(v, dict) and k.startswith("c3"):
        print(stage, k, round(v["ms_per_step"], 3), "serial", round(v["serial_calls"]["ms_per_step"], 3))
m.close()
