"""The C ABI's PCM gather (include/vits.h vits_pcm_gather_*) on ONE GPU: world 1 without RCCL, and — VITS_GATHER_FORCE_RCCL=1 — through a real
one-rank RCCL communicator (ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy of the N > 1 path), fp32 and PCM16, against the
rows vits_model_process_batch returns on the host. No torch: device buffers through libamdhip64 directly, as a C host would.
Run by tests/test_gpu_edge_and_scale.py in its own process. Prints `gather_check ok`."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package

pkg = load_package()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n) == 0
    assert hip.hipMemset(p, 0xFF, n) == 0
    return p.value


def to_host(ptr, shape, dtype):
    out = np.empty(shape, dtype)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY))
lens_ids = np.array([12, 3, 9, 1, 7], np.int32)
ids = np.zeros((5, 12), np.int32)
for b, n in enumerate(lens_ids):
    ids[b, :n] = pkg.synth_ids(1, int(n), ids_seed=50 + b)[0]
want, lengths, _ = m.process_batch(ids, id_lengths=lens_ids, noise_seed=9)
cap = int(lengths.max()) + 37
pcm = dmalloc(5 * cap * 4)
m.process_batch(ids, id_lengths=lens_ids, noise_seed=9, out_device=pcm, out_device_stride=cap, skip_host_copy=True, keep_pcm=False)
pcm16 = dmalloc(5 * cap * 2)
lens_dev = dmalloc(5 * 8)
assert hip.hipMemcpy(lens_dev, lengths.astype(np.int64).ctypes.data_as(C.c_void_p), 40, 1) == 0
pkg.pcm16_device(pcm, cap, pcm16, cap, 5, cap, lengths_ptr=lens_dev)
m.sync()
assert hip.hipDeviceSynchronize() == 0

for forced in (False, True):
    if forced:
        os.environ["VITS_GATHER_FORCE_RCCL"] = "1"
    uid = None
    for eb, src, dtype in ((4, pcm, np.float32), (2, pcm16, np.int16)):
        if forced:
            uid = pkg.gather_unique_id()  # (one id per communicator: RCCL's bootstrap serves an id once)
            assert len(uid) == 128 and any(uid)
        with pkg.PcmGather(uid, 0, 1, 5, cap, eb) as g:
            for rep in range(2):  # the object is reusable
                data, stride, all_len = g.gather(src, cap, lengths)
                assert stride == int(lengths.max()) and np.array_equal(all_len, lengths)
                host = to_host(data, (5, stride), dtype)
                for b in range(5):
                    ref = want[b] if eb == 4 else pkg.pcm16(want[b])
                    assert np.array_equal(host[b, : lengths[b]], ref), (forced, eb, b)
            # misuse is refused with a message
            for bad in (lengths + cap, -lengths):
                try:
                    g.gather(src, cap, bad)
                    raise SystemExit("a bad length was accepted")
                except pkg.VitsError as e:
                    assert "longer than" in str(e), e
    try:
        pkg.PcmGather(uid, 1, 1, 5, cap, 4)
        raise SystemExit("rank >= world was accepted")
    except pkg.VitsError:
        pass
m.close()
for p in (pcm, pcm16, lens_dev):
    hip.hipFree(p)
print("gather_check ok")
