"""One rank of the TWO-rank run of the C ABI's PCM gather (include/vits.h vits_pcm_gather_*) on ONE GPU: two processes, RCCL replaced by the
shared-directory test transport tools/stub_rccl_shm.c (RCCL refuses two ranks on one device; this pool has one GPU per box). What runs is the
product's whole N > 1 path — capacity + lengths all-gather, the collective verdict, rank-block placement, rows padded to the longest utterance of
ANY rank, fp32 and PCM16 — and its failure behaviour: a rank with unusable arguments makes EVERY rank return -1, nobody hangs,
the object stays usable. usage: gather2_rank.py RANK WORLD ID_FILE ; env VITS_RCCL_LIB, STUB_RCCL_DIR. Started by tests/test_gpu_gather2.py."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package

rank, world, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pkg = load_package()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n) == 0
    assert hip.hipMemset(p, 0xFF, n) == 0
    return p.value


def to_host(ptr, shape, dtype):
    out = np.empty(shape, dtype)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, 2) == 0
    return out


_n_ids = [0]


def fresh_uid():
    """one unique id per communicator (RCCL's bootstrap serves an id once); the 128 bytes travel through a file (the host application's job)"""
    path = "%s.%d" % (id_file, _n_ids[0])
    _n_ids[0] += 1
    if rank == 0:
        uid = pkg.gather_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 60, "rank 0 never wrote the unique id"
        time.sleep(0.01)
    return open(path, "rb").read()


m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY))
rows = 3
id_lens = np.array([12, 3, 9, 5, 1, 6], np.int32)
ids = np.zeros((6, 12), np.int32)
for b, n in enumerate(id_lens):
    ids[b, :n] = pkg.synth_ids(1, int(n), ids_seed=70 + b)[0]
glob = np.arange(6, dtype=np.int32)
_, l0, _ = m.process_batch(ids, id_lengths=id_lens, noise_seed=9, noise_seed_offsets=glob, keep_pcm=False)
order = np.argsort(-l0.astype(np.int64), kind="stable")  # rank 0 owns the three LONGEST utterances: the ranks' local maxima differ
ids, id_lens, glob = ids[order], id_lens[order], glob[order].astype(np.int32)  # (an utterance's noise stream follows it: noise_seed_offsets)
want, lengths, _ = m.process_batch(ids, id_lengths=id_lens, noise_seed=9, noise_seed_offsets=glob)  # every rank knows the whole answer
lengths = lengths.astype(np.int64)
mine = slice(rank * rows, (rank + 1) * rows)
own_max, smax = int(lengths[mine].max()), int(lengths.max())
assert int(lengths[:3].max()) > int(lengths[3:].max()) + 8, lengths
cap = smax + 37
stride = cap if rank == 0 else own_max + 5  # rank 1's buffer is NARROWER than the common row width: only its own rows must fit
assert rank == 0 or stride < smax
pcm = dmalloc(rows * stride * 4)
m.process_batch(ids[mine], id_lengths=id_lens[mine], noise_seed=9, noise_seed_offsets=glob[mine], out_device=pcm, out_device_stride=stride, skip_host_copy=True, keep_pcm=False)
pcm16 = dmalloc(rows * stride * 2)
lens_dev = dmalloc(rows * 8)
assert hip.hipMemcpy(lens_dev, lengths[mine].copy().ctypes.data_as(C.c_void_p), rows * 8, 1) == 0
pkg.pcm16_device(pcm, stride, pcm16, stride, rows, stride, lengths_ptr=lens_dev)
m.sync()
assert hip.hipDeviceSynchronize() == 0


def expect_failure(g, src, st, lens, needle):
    t0 = time.time()
    try:
        g.gather(src, st, lens)
    except pkg.VitsError as e:
        assert needle in str(e), (needle, str(e))
        assert time.time() - t0 < 15, "the failing exchange took %.1f s" % (time.time() - t0)
        return
    raise SystemExit("rank %d: a failing exchange succeeded (%s)" % (rank, needle))


for eb, src, dtype in ((4, pcm, np.float32), (2, pcm16, np.int16)):
    with pkg.PcmGather(fresh_uid(), rank, world, rows, cap, eb) as g:
        def good():
            data, st, all_len = g.gather(src, stride, lengths[mine])
            assert st == smax and np.array_equal(all_len, lengths), (st, smax, all_len, lengths)
            host = to_host(data, (world * rows, st), dtype)
            for b in range(world * rows):  # every row of EVERY rank, at its place in rank order
                ref = want[b] if eb == 4 else pkg.pcm16(want[b])
                assert np.array_equal(host[b, : lengths[b]], ref), (rank, eb, b)
        good()
        # (A) rank 1 passes a stride shorter than its own longest row; rank 0's arguments are fine: BOTH return -1 naming rank 1
        if rank == 1:
            expect_failure(g, src, own_max - 1, lengths[mine], "rank 1")
        else:
            expect_failure(g, src, stride, lengths[mine], "rank 1")
        # (B) rank 0 claims a row longer than the agreed capacity: both return -1 naming rank 0
        bad = lengths[mine].copy()
        bad[1] = cap + 1
        expect_failure(g, src, stride, bad if rank == 0 else lengths[mine], "rank 0")
        # (C) rank 1 passes no buffer at all
        if rank == 1:
            expect_failure(g, None, stride, lengths[mine], "rank 1")
        else:
            expect_failure(g, src, stride, lengths[mine], "rank 1")
        good()  # the object is still usable: failures were collective, the two ranks are in step
    # (D) ranks that disagree on row_capacity are told so, together
    with pkg.PcmGather(fresh_uid(), rank, world, rows, cap + rank, eb) as g2:
        expect_failure(g2, src, stride, lengths[mine], "row_capacity")
m.close()
print("gather2 ok rank %d" % rank)
