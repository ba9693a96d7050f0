"""world_size-2 test of the N>1 path on CPU (gloo): utterance sharding and the ragged all-gather of PCM that bench.py
uses over RCCL. No GPU, no product compute: the PCM rows are synthetic."""
import importlib.util
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_multi_gpu():
    spec = importlib.util.spec_from_file_location("vits_multi_gpu", os.path.join(ROOT, "vits.cpp_amd", "multi_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_pcm(utt, length, cap, dtype="float32"):
    if dtype == "int16":  # what mg.to_pcm16 hands to the gather on the GPU (half the bytes)
        row = torch.zeros(cap, dtype=torch.int16)
        row[:length] = ((torch.arange(length) * 7 + 1000 * utt) % 65536 - 32768).to(torch.int16)
        return row
    row = torch.zeros(cap)
    row[:length] = torch.arange(length, dtype=torch.float32) * 1e-3 + utt
    return row


def _worker(rank, world, port, total, cap, q, dtype="float32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = _load_multi_gpu()
    lo, hi = mg.shard_range(total, world, rank)
    lengths = torch.tensor([100 + 37 * u for u in range(lo, hi)], dtype=torch.int64)
    pcm = torch.stack([_fake_pcm(u, int(lengths[i]), cap, dtype) for i, u in enumerate(range(lo, hi))])
    out, all_len = mg.gather_pcm(pcm, lengths)
    q.put((rank, out.numpy().copy(), all_len.numpy().copy()))  # numpy: pickled by value (torch tensors travel as fds)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    mg = _load_multi_gpu()
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [mg.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_shards_equalise_the_frame_sums():
    """SURVEY 8e "balance by sum of predicted frames": same counts per rank as the contiguous split, smaller spread of the
    per-rank sums, every utterance assigned exactly once, identical on every rank (pure function of the weights)."""
    import numpy as np
    mg = _load_multi_gpu()
    rng = np.random.default_rng(3)
    for world, total in ((8, 512), (2, 7), (3, 10), (4, 64)):
        frames = rng.integers(150, 320, size=total)
        frames[:: max(total // 5, 1)] *= 3  # a few long outliers
        shards = mg.balanced_shards(frames, world)
        assert sorted(i for s in shards for i in s) == list(range(total))
        assert [len(s) for s in shards] == [mg.shard_range(total, world, r)[1] - mg.shard_range(total, world, r)[0] for r in range(world)]
        contiguous = [list(range(*mg.shard_range(total, world, r))) for r in range(world)]
        assert mg.imbalance(frames, shards) <= mg.imbalance(frames, contiguous) + 1e-12
        assert shards == mg.balanced_shards(list(frames), world)
    # the benchmark's own situation: 8 ranks x 64 utterances of ~225 +- 25 frames
    frames = rng.normal(225, 25, size=512).astype(int)
    assert mg.imbalance(frames, mg.balanced_shards(frames, 8)) < 1.001


def _worker_uneven(rank, world, port, total, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = _load_multi_gpu()
    lo, hi = mg.shard_range(total, world, rank)
    lengths = torch.tensor([100 + 37 * u for u in range(lo, hi)], dtype=torch.int64)
    pcm = torch.stack([_fake_pcm(u, int(lengths[i]), cap) for i, u in enumerate(range(lo, hi))])
    out, all_len = mg.gather_pcm(pcm, lengths)
    frames_all = mg.gather_frames(lengths // 10)
    q.put((rank, out.numpy().copy(), all_len.numpy().copy(), frames_all.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_uneven_shards_all_gather_world2():
    """shard_range hands out blocks that differ by one when total % world != 0; the fixed-size collective then
    needs the short block padded (and the padding dropped again). 7 utterances over 2 ranks: 4 + 3."""
    world, total, cap = 2, 7, 512
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, total, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    want_len = torch.tensor([100 + 37 * u for u in range(total)], dtype=torch.int64)
    for rank, out, all_len, frames_all in results:
        out, all_len = torch.from_numpy(out), torch.from_numpy(all_len)
        assert torch.equal(all_len, want_len) and out.shape == (total, int(want_len.max()))
        assert torch.equal(torch.from_numpy(frames_all), want_len // 10)
        for u in range(total):
            assert torch.equal(out[u, : want_len[u]], _fake_pcm(u, int(want_len[u]), cap)[: want_len[u]])


def test_bench_launcher_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus 8` without torchrun starts the ranks itself, and fails (exit 2, message) when fewer devices
    are visible — it must never fall back to one rank and print an n_gpus=1 line labelled as a scaling run.
    Here (no GPU) 0 devices are visible."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=110)
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 devices are visible here")
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr and not r.stdout.strip()
    # a rank process whose --gpus disagrees with the launcher's world size is an error as well
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 2 and "does not match WORLD_SIZE" in r.stderr


@pytest.mark.timeout(120)
@pytest.mark.parametrize("dtype", ["float32", "int16"])
def test_ragged_pcm_all_gather_world2(dtype):
    world, total, cap = 2, 8, 1024
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, cap, q, dtype)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    want_len = torch.tensor([100 + 37 * u for u in range(total)], dtype=torch.int64)
    smax = int(want_len.max())
    for rank, out, all_len in results:
        out, all_len = torch.from_numpy(out), torch.from_numpy(all_len)
        assert torch.equal(all_len, want_len)
        assert out.shape == (total, smax) and str(out.dtype) == "torch." + dtype
        for u in range(total):
            assert torch.equal(out[u, : want_len[u]], _fake_pcm(u, int(want_len[u]), cap, dtype)[: want_len[u]])
            assert float(out[u, want_len[u]:].float().abs().sum()) == 0.0


# ---- world 8 (the node size of BASELINE.json configs[3]) and the pipelined exchange ----------------------------------------------
def _spawn(target, world, args, timeout=150):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    return sorted(results, key=lambda r: r[0])


def _len_of(u, step=0):
    return 100 + 37 * ((u * 5 + step * 3) % 11)


def _worker_world8(rank, world, port, q, total, cap, balanced):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mg = _load_multi_gpu()
    weights = [_len_of(u) for u in range(total)]
    shards = mg.balanced_shards(weights, world) if balanced else [list(range(*mg.shard_range(total, world, r))) for r in range(world)]
    mine = shards[rank]
    lengths = torch.tensor([_len_of(u) for u in mine], dtype=torch.int64)
    pcm = torch.stack([_fake_pcm(u, int(lengths[i]), cap) for i, u in enumerate(mine)]) if mine else torch.zeros((0, cap))
    out, all_len = mg.gather_pcm(pcm, lengths)
    out, all_len = mg.restore_order(out, all_len, shards)
    fr = mg.gather_frames(lengths // 10)
    q.put((rank, out.numpy().copy(), all_len.numpy().copy(), fr.numpy().copy(), mg.gather_index_map(shards)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(200)
@pytest.mark.parametrize("total,balanced", [(19, False), (19, True), (5, False)], ids=["uneven", "balanced", "empty_shards"])
def test_world8_gather_uneven_balanced_and_empty_shards(total, balanced):
    """Eight ranks: 19 utterances (blocks of 3 and 2), the same with frame-balanced (non-contiguous) shards put back into global
    order with the returned index map, and 5 utterances (three ranks with an EMPTY shard: ADVICE r2 — they must send an
    all-padding block instead of crashing in lengths.max() while the others hang in the collective)."""
    world, cap = 8, 600
    results = _spawn(_worker_world8, world, (total, cap, balanced))
    want_len = torch.tensor([_len_of(u) for u in range(total)], dtype=torch.int64)
    for rank, out, all_len, fr, imap in results:
        out, all_len = torch.from_numpy(out), torch.from_numpy(all_len)
        assert sorted(imap) == list(range(total))
        assert torch.equal(all_len, want_len) and out.shape == (total, int(want_len.max()))
        assert sorted(fr.tolist()) == sorted((want_len // 10).tolist())
        for u in range(total):
            assert torch.equal(out[u, : want_len[u]], _fake_pcm(u, int(want_len[u]), cap)[: want_len[u]])


def _worker_exchange(rank, world, port, q, total, cap, steps, dtype):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mg = _load_multi_gpu()
    lo, hi = mg.shard_range(total, world, rank)
    blocks = []
    tdt = torch.int16 if dtype == "int16" else torch.float32
    ex = mg.PcmExchange(hi - lo, cap, dtype=tdt, device="cpu", on_block=lambda step, out, lens: blocks.append((step, out.numpy().copy(), lens.numpy().copy())))
    allocs = (ex.out.data_ptr(), ex.send.data_ptr())
    bufs = [torch.zeros((hi - lo, cap), dtype=tdt) for _ in range(3)]
    for i in range(steps):
        buf = bufs[i % 3]
        lengths = torch.tensor([_len_of(u, i) for u in range(lo, hi)], dtype=torch.int64)
        for r, u in enumerate(range(lo, hi)):
            buf[r] = _fake_pcm(u + 1000 * i, int(lengths[r]), cap, dtype)
        ex.submit(buf, lengths)
        assert len(blocks) == i  # step i's block is produced one submit later: it never delays the step that made it
    ex.flush()
    assert (ex.out.data_ptr(), ex.send.data_ptr()) == allocs  # nothing was reallocated
    q.put((rank, blocks))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(200)
@pytest.mark.parametrize("world,total,dtype", [(8, 19, "float32"), (8, 64, "int16"), (2, 3, "float32"), (8, 5, "float32")])
def test_pipelined_exchange_delivers_every_step_in_order(world, total, dtype):
    """PcmExchange: no meta exchange, no .tolist() round trip and no allocation per step — the lengths of step i travel with step
    i, the PCM of step i is gathered (exact common width) while step i + 1 is produced, flush() drains. Every step's block must
    arrive once, in order, with every row equal to what its owner wrote — uneven shards, empty shards, fp32 and int16 rows."""
    cap, steps = 600, 5
    results = _spawn(_worker_exchange, world, (total, cap, steps, dtype))
    for rank, blocks in results:
        assert [b[0] for b in blocks] == list(range(steps))
        for step, out, lens in blocks:
            want_len = [_len_of(u, step) for u in range(total)]
            assert lens.tolist() == want_len and out.shape == (total, max(want_len))
            for u in range(total):
                row = _fake_pcm(u + 1000 * step, want_len[u], cap, dtype).numpy()
                assert (out[u, : want_len[u]] == row[: want_len[u]]).all() and not out[u, want_len[u]:].any()


# ---- the C ABI's gather (include/vits.h vits_pcm_gather_*): unique-id plumbing with RCCL stubbed out --------------------------------
_STUB_RCCL = r'''
/* test stub of the five RCCL entry points pcm_gather.cpp binds (rccl.h:187,220,260,339,678): records what it is called with */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
typedef struct { char internal[128]; } ncclUniqueId;
static void logline(const char* what, const ncclUniqueId* id, int rank, int world) {
    const char* path = getenv("STUB_RCCL_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%s %d %d ", what, rank, world);
    if (id) for (int i = 0; i < 128; ++i) fprintf(f, "%02x", (unsigned char)id->internal[i]);
    fprintf(f, "\n");
    fclose(f);
}
int ncclGetUniqueId(ncclUniqueId* id) {
    for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 7 + 3 + (getpid() & 0x3f));
    logline("id", id, -1, -1);
    return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) { *comm = malloc(8); logline("init", &id, rank, nranks); return 0; }
int ncclCommDestroy(void* comm) { free(comm); logline("destroy", NULL, -1, -1); return 0; }
const char* ncclGetErrorString(int r) { (void)r; return "stub error"; }
int ncclAllGather(const void* s, void* r, size_t n, int dt, void* c, void* st) { (void)s; (void)r; (void)n; (void)dt; (void)c; (void)st; return 0; }
'''


def _gather_id_worker(rank, world, port, stub, log, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "VITS_RCCL_LIB": stub, "STUB_RCCL_LOG": log})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    pkg = load_package()
    # rank 0 creates the id through the C ABI; the HOST APPLICATION moves the 128 bytes (here: a gloo broadcast)
    buf = torch.zeros(pkg.GATHER_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(pkg.gather_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(buf, 0)
    uid = bytes(buf.tolist())
    said = []
    for bad in (uid[:64], None):  # a truncated / missing id is refused before RCCL sees it
        try:
            pkg.PcmGather(bad, rank, world, 4, 1000)
            said.append("accepted")
        except pkg.VitsError as e:
            said.append(str(e))
    try:
        pkg.PcmGather(uid, rank, world, 4, 1000)  # joins (stub), then needs a device for its buffers: there is none here
        said.append("accepted")
    except pkg.VitsError as e:
        said.append(str(e))
    q.put((rank, uid.hex(), said))
    dist.barrier()
    dist.destroy_process_group()


def test_c_abi_gather_unique_id_reaches_every_rank(tmp_path):
    """vits_pcm_gather_unique_id / vits_pcm_gather_init under two gloo-launched CPU processes with RCCL replaced by a recording stub
    (VITS_RCCL_LIB): rank 0's 128 bytes arrive unchanged in ncclCommInitRank on both ranks with the right (rank, world); a malformed id
    never reaches RCCL; and with no GPU the object refuses to exist (no CPU path) after giving the communicator back."""
    import subprocess
    src = tmp_path / "stub_rccl.c"
    src.write_text(_STUB_RCCL)
    stub = str(tmp_path / "libstub_rccl.so")
    cc = subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", str(src), "-o", stub], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    log = str(tmp_path / "stub.log")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_id_worker, args=(r, 2, port, stub, log, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and len(got[0][1]) == 256
    for rank, uid, said in got:
        assert "128 bytes" in said[0] and "128 bytes" in said[1], said
        assert "accepted" not in said[2] and ("device" in said[2].lower() or "hip" in said[2].lower()), said
    lines = [ln.split() for ln in open(log).read().splitlines()]
    inits = sorted((int(ln[1]), int(ln[2]), ln[3]) for ln in lines if ln[0] == "init")
    assert inits == [(0, 2, got[0][1]), (1, 2, got[0][1])]
    assert sum(ln[0] == "id" for ln in lines) == 1 and sum(ln[0] == "destroy" for ln in lines) == 2


def test_gather_verdict_is_one_decision_for_every_rank():
    """vits_pcm_gather_verdict (csrc/pcm_gather.cpp PcmGather::verdict): the decision vits_pcm_gather takes on the table of its first all-gather
    — per rank [row_capacity, lengths...], -1 for a row its rank could not use — is a pure function of that table, so every rank reaches it
    together (a rank-local early return left the peers blocked in ncclAllGather). Good tables give the common row width;
    a -1 row, an over-long row and disagreeing capacities are refused with the (rank, row) in the message."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    pkg = load_package()
    assert pkg.gather_verdict([[100, 5, 0, 99], [100, 100, 7, 1]], 2, 3) == 100
    assert pkg.gather_verdict([[100, 0, 0, 0]], 1, 3) == 1  # (an all-empty shard still has a row width)
    for table, needle in (([[100, 5, 0, 99], [100, 3, -1, 1]], "rank 1 passed an unusable row 1"),
                          ([[100, -1, -1, -1], [100, 3, 2, 1]], "rank 0 passed an unusable row 0"),
                          ([[100, 5, 101, 99], [100, 3, 2, 1]], "row 1 of rank 0 is longer than row_capacity"),
                          ([[100, 5, 1, 99], [101, 3, 2, 1]], "rank 1 was initialised with row_capacity 101")):
        with pytest.raises(pkg.VitsError) as e:
            pkg.gather_verdict(table, 2, 3)
        assert needle in str(e.value), (needle, str(e.value))
