"""One rank of the multi-GPU parity check (tests/test_gpu_round2.py::test_rccl_gather_*): every rank synthesises its shard of
a small global batch on its own GPU, the PCM is all-gathered over RCCL (torch.distributed backend "nccl"), and every rank
verifies the gathered tensor row by row: its own rows bit-equal to what it computed, and every OTHER rank's rows bit-equal
to a local recomputation of those utterances (utterances are deterministic functions of (ids, seed), wherever they run).
Also exercises the frames-balanced dispatch (frames_only pre-pass -> gather_frames -> balanced_shards -> noise_seed_offsets).
Started with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; prints `rank R ok`."""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pkg = load("vits_cpp_amd", os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
    mg = load("vits_multi_gpu", os.path.join(ROOT, "vits.cpp_amd", "multi_gpu.py"))
    pkg.set_device(local)
    total, T, base = 5 * world + (1 if world > 1 else 0), 24, 777  # uneven blocks when world > 1
    ids_all = pkg.synth_ids(total, T)
    ids_all[1::2, 12:] = 0  # (all lengths equal; content differs per utterance)
    m = pkg.Model(pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL))
    cap = 256 * 8 * T + 294

    def run(indices):
        out = torch.zeros((len(indices), cap), dtype=torch.float32, device="cuda")
        _, lengths, frames = m.process_batch(ids_all[indices], noise_seed=base, noise_seed_offsets=np.asarray(indices, np.int32), out_device=out.data_ptr(),
                                             out_device_stride=cap, skip_host_copy=True)
        return out, torch.from_numpy(lengths).cuda(), frames

    # 1) contiguous shards -------------------------------------------------------------------------------------------
    lo, hi = mg.shard_range(total, world, rank)
    mine = list(range(lo, hi))
    out, lens, frames = run(mine)
    for pcm16 in (False, True):
        g, gl = mg.gather_pcm(mg.to_pcm16(pkg, out, lens) if pcm16 else out, lens)
        assert g.shape[0] == total and gl.shape[0] == total, (g.shape, gl.shape)
        ref_all, ref_len, _ = run(list(range(total)))  # every utterance recomputed locally
        if pcm16:
            ref_all = mg.to_pcm16(pkg, ref_all, ref_len)
        assert torch.equal(gl, ref_len)
        for u in range(total):
            n = int(ref_len[u])
            assert torch.equal(g[u, :n], ref_all[u, :n]), f"rank {rank}: utterance {u} differs after the gather (pcm16={pcm16})"
            assert not bool(g[u, n:].any())
    # 2) frames-balanced shards -----------------------------------------------------------------------------------------
    _, _, fr = m.process_batch(ids_all[mine], noise_seed=base, noise_seed_offsets=np.asarray(mine, np.int32), frames_only=True)
    assert np.array_equal(fr, frames)  # the pre-pass predicts exactly the frames the full call produces
    all_fr = mg.gather_frames(torch.from_numpy(fr).cuda()).cpu().numpy()
    assert all_fr.size == total
    shards = mg.balanced_shards(all_fr, world)
    out_b, lens_b, _ = run(shards[rank])
    g, gl = mg.gather_pcm(out_b, lens_b)
    order = [u for s in shards for u in s]
    ref_all, ref_len, _ = run(list(range(total)))
    for row, u in enumerate(order):
        n = int(ref_len[u])
        assert int(gl[row]) == n and torch.equal(g[row, :n], ref_all[u, :n]), f"rank {rank}: balanced row {row} (utterance {u}) differs"
    # 3) the pipelined exchange bench.py uses (multi_gpu.PcmExchange): three steps with different noise seeds through three rotating
    # buffers, fp32 and int16; every block must arrive once, in order, every row bit-equal to a local recomputation ---------------
    for pcm16 in (False, True):
        blocks = []
        ex = mg.PcmExchange(len(mine), cap, dtype=torch.int16 if pcm16 else torch.float32, device="cuda",
                            on_block=lambda step, g_, gl_: blocks.append((step, g_.clone(), gl_.clone())))
        bufs = [torch.zeros((len(mine), cap), dtype=torch.float32, device="cuda") for _ in range(3)]
        bufs16 = [torch.zeros((len(mine), cap), dtype=torch.int16, device="cuda") for _ in range(3)]
        for i in range(3):
            _, lengths, _ = m.process_batch(ids_all[mine], noise_seed=base + 100 * i, noise_seed_offsets=np.asarray(mine, np.int32), out_device=bufs[i].data_ptr(),
                                            out_device_stride=cap, skip_host_copy=True)
            lens_d = torch.from_numpy(lengths).cuda()
            src = bufs[i]
            if pcm16:
                pkg.pcm16_device(src.data_ptr(), src.stride(0), bufs16[i].data_ptr(), bufs16[i].stride(0), len(mine), cap, lengths_ptr=lens_d.data_ptr(),
                                 stream=torch.cuda.current_stream().cuda_stream)
                src = bufs16[i]
            ex.submit(src, lens_d)
            assert len(blocks) == i
        ex.flush()
        assert [b[0] for b in blocks] == [0, 1, 2]
        for i, (_, g, gl) in enumerate(blocks):
            ref = torch.zeros((total, cap), dtype=torch.float32, device="cuda")
            _, rl, _ = m.process_batch(ids_all, noise_seed=base + 100 * i, noise_seed_offsets=np.arange(total, dtype=np.int32), out_device=ref.data_ptr(), out_device_stride=cap,
                                       skip_host_copy=True)
            rl = torch.from_numpy(rl)
            if pcm16:
                ref = mg.to_pcm16(pkg, ref, rl.cuda())
            assert torch.equal(gl.cpu(), rl), (i, pcm16)
            for u in range(total):
                n = int(rl[u])
                assert torch.equal(g[u, :n], ref[u, :n]), f"rank {rank}: exchange step {i} utterance {u} differs (pcm16={pcm16})"
    # 4) padded blocks on the exchange stream: row_capacity > rows makes every rank's block carry padding rows, so the
    # `keep` path runs even with one rank. The exchange's buffers are POISONED first: a reader that is not ordered behind the
    # all-gather (or an index_select on another stream) would hand out poison, not a lucky copy of the right data -------------------
    for pcm16 in (False, True):
        blocks = []
        ex = mg.PcmExchange(len(mine), cap, dtype=torch.int16 if pcm16 else torch.float32, device="cuda", row_capacity=len(mine) + 2,
                            on_block=lambda step, g_, gl_: blocks.append((step, g_.clone(), gl_.clone())))
        assert ex.keep is not None and ex.kept is not None
        poison = 0x7A7A if pcm16 else float("nan")
        ex.out.fill_(poison)
        ex.kept.fill_(poison)
        ex.send.fill_(poison)
        torch.cuda.synchronize()
        bufs = [torch.zeros((len(mine), cap), dtype=torch.float32, device="cuda") for _ in range(3)]
        bufs16 = [torch.zeros((len(mine), cap), dtype=torch.int16, device="cuda") for _ in range(3)]
        refs = []
        for i in range(4):
            _, lengths, _ = m.process_batch(ids_all[mine], noise_seed=base + 7 * i, noise_seed_offsets=np.asarray(mine, np.int32), out_device=bufs[i % 3].data_ptr(),
                                            out_device_stride=cap, skip_host_copy=True)
            lens_d = torch.from_numpy(lengths).cuda()
            src = bufs[i % 3]
            if pcm16:
                pkg.pcm16_device(src.data_ptr(), src.stride(0), bufs16[i % 3].data_ptr(), bufs16[i % 3].stride(0), len(mine), cap, lengths_ptr=lens_d.data_ptr(),
                                 stream=torch.cuda.current_stream().cuda_stream)
                src = bufs16[i % 3]
            refs.append((src.clone(), lengths.copy()))
            ex.submit(src, lens_d)
        ex.flush()
        assert [b[0] for b in blocks] == [0, 1, 2, 3] and ex.bmax == len(mine) + 2
        lo_row = sum(ex.counts[:rank])
        for i, (_, g, gl) in enumerate(blocks):
            assert g.shape[0] == total and gl.shape[0] == total  # padding rows gone
            want, wl = refs[i]
            for r in range(len(mine)):
                n = int(wl[r])
                assert int(gl[lo_row + r]) == n
                assert torch.equal(g[lo_row + r, :n], want[r, :n]), f"rank {rank}: padded exchange step {i} row {r} differs (pcm16={pcm16})"
    dist.barrier()
    dist.destroy_process_group()
    m.close()
    print(f"rank {rank} ok world {world} imbalance contiguous {mg.imbalance(all_fr, [list(range(*mg.shard_range(total, world, r))) for r in range(world)]):.4f} "
          f"balanced {mg.imbalance(all_fr, shards):.4f}", flush=True)


if __name__ == "__main__":
    main()
