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
# An opts.async call (returns with its flow / vocoder still queued on the main stream) followed by a pipelined submit on the same handle:
# stage one of the submit runs on the front-end stream into the SAME stage-one arena (slot 0) the async batch is still reading; the engine
# orders it behind the async tail. Different ids in the two batches, so an overwritten arena would change the async batch's PCM.
m.set_arith(pkg.ARITH_F32)
big, other = pkg.synth_ids(24, 128), pkg.synth_ids(24, 128, ids_seed=777)
cap = 256 * 2 * 128 + 294
kw = dict(fixed_duration=2, out_device_stride=cap, skip_host_copy=True)
ba, bb = (torch.zeros((24, cap), dtype=torch.float32, device="cuda") for _ in range(2))
want_a = m.process_batch(big, noise_seed=11, fixed_duration=2)
want_b = m.process_batch(other, noise_seed=12, fixed_duration=2)
for _ in range(3):
    ba.zero_(); bb.zero_(); torch.cuda.synchronize()
    m.process_batch(big, noise_seed=11, out_device=ba.data_ptr(), async_=True, keep_pcm=False, **kw)
    m.submit_batch(other, noise_seed=12, out_device=bb.data_ptr(), **kw)
    m.wait()
    m.sync()
    ha, hb = ba.cpu().numpy(), bb.cpu().numpy()
    for b in range(24):
        assert np.array_equal(ha[b, : want_a[1][b]], want_a[0][b]), ("async batch corrupted by the following submit", b)
        assert np.array_equal(hb[b, : want_b[1][b]], want_b[0][b]), ("submit after async", b)
m.close()
print("pipe_check ok")
