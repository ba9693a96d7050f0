"""Pipelined batches writing to caller-owned device buffers (vits_process_opts.out_device), with predicted durations, pinned durations
(no host read between the stages: an event orders the two streams) and the windowed vocoder: PCM, lengths and frames must equal the
serial vits_model_process_batch results bit for bit. Run by tests/test_gpu_pipeline.py in its own process. Prints `pipe_check ok`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
import torch

torch.zeros(1, device="cuda")  # torch's HIP runtime first (as in bench.py)
pkg = load_package()
m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
ids = pkg.synth_ids(4, 48)
cap = 256 * 8 * 48 + 294
for arith in (pkg.ARITH_F32, pkg.ARITH_F16):
    m.set_arith(arith)
    for kw in (dict(), dict(fixed_duration=2), dict(vocoder_chunk_frames=37)):
        want = [m.process_batch(ids, noise_seed=70 + s, **kw) for s in range(3)]
        bufs = [torch.zeros((4, cap), dtype=torch.float32, device="cuda") for _ in range(3)]
        m.submit_batch(ids, noise_seed=70, out_device=bufs[0].data_ptr(), out_device_stride=cap, skip_host_copy=True, **kw)
        res = []
        for s in (1, 2):
            m.submit_batch(ids, noise_seed=70 + s, out_device=bufs[s].data_ptr(), out_device_stride=cap, skip_host_copy=True, **kw)
            res.append(m.wait())
        res.append(m.wait())
        for s in range(3):
            pw, lw, fw = want[s]
            pg, lg, fg = res[s]
            assert pg is None and np.array_equal(lw, lg) and np.array_equal(fw, fg), (arith, kw, s)
            host = bufs[s].cpu().numpy()
            for b in range(4):
                assert np.array_equal(host[b, : lw[b]], pw[b]), (arith, kw, s, b)
m.close()
print("pipe_check ok")
