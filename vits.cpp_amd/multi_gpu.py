"""Multi-GPU plumbing: one process per GPU, utterances sharded across ranks, ONE exchange — the all-gather of the PCM.

The reference processes exactly one utterance per call (src/vits.cpp:184,303), so utterances are independent: every rank
synthesises its own shard with replicated weights and no data-path collective. The only exchange is the final gather of
the PCM (ragged: per-utterance lengths travel first), over torch.distributed (backend "nccl" == RCCL on ROCm, xGMI between
the 8 GPUs of a node; "gloo" on CPU for the tests). Payload at the benchmark shape: 64 x ~58k x 4 B ~ 15 MB per rank —
negligible next to the >80 ms of compute per step, so a plain all_gather_into_tensor is used.

Sharding: `shard_range` = contiguous blocks; `balanced_shards` = blocks of equal COUNT whose summed
weights (predicted frames from a `frames_only` call, or id counts as a free proxy) are as equal as a greedy
longest-first assignment makes them — the slowest rank sets the step time, and the vocoder's work is linear in frames.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block of utterances owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def balanced_shards(weights, world):
    """Assign utterance i (cost weights[i], e.g. its predicted frame count) to one of `world` ranks so that the per-rank sums
    are balanced while the per-rank COUNTS stay those of `shard_range` (weak scaling keeps its batch per GPU).
    Greedy longest-processing-time: utterances in descending weight go to the least-loaded rank that still has room;
    ties broken by the lower rank / lower index, so every rank computes the same assignment from the same weights.
    Returns a list of `world` index lists (each sorted ascending)."""
    n = len(weights)
    cap = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
    order = sorted(range(n), key=lambda i: (-int(weights[i]), i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min((r for r in range(world) if len(shards[r]) < cap[r]), key=lambda r: (load[r], r))
        shards[r].append(i)
        load[r] += int(weights[i])
    return [sorted(s) for s in shards]


def imbalance(weights, shards):
    """max over ranks of the summed weight / mean over ranks (1.0 = perfectly balanced): the factor by which the slowest
    rank stretches a step whose cost is linear in the weights."""
    sums = [sum(int(weights[i]) for i in s) for s in shards]
    mean = sum(sums) / max(len(sums), 1)
    return max(sums) / mean if mean > 0 else 1.0


def to_pcm16(pkg, pcm, lengths=None):
    """fp32 PCM [B, cap] on the GPU -> int16 [B, cap] with the library's device kernel (vits_pcm16_from_float_device, the
    reference driver's conversion test/main.cpp:31-33), on torch's current stream. Halves the bytes gather_pcm moves."""
    assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.stride(1) == 1
    out = torch.zeros(pcm.shape, dtype=torch.int16, device=pcm.device)
    lp = None
    if lengths is not None:
        lengths = lengths.to(device=pcm.device, dtype=torch.int64).contiguous()
        lp = lengths.data_ptr()
    pkg.pcm16_device(pcm.data_ptr(), pcm.stride(0), out.data_ptr(), out.stride(0), pcm.shape[0], pcm.shape[1], lengths_ptr=lp,
                     stream=torch.cuda.current_stream().cuda_stream)
    return out


def gather_frames(frames):
    """frames: [B] int64 tensor of this rank (predicted frames of its contiguous block). Returns the concatenation over
    ranks (a control message of 8 bytes per utterance, not part of the data path). Blocks may differ in size by one."""
    if not dist.is_available() or not dist.is_initialized():
        return frames
    world = dist.get_world_size()
    n = torch.tensor([frames.numel()], dtype=torch.int64, device=frames.device)
    counts = torch.empty(world, dtype=torch.int64, device=frames.device)
    dist.all_gather_into_tensor(counts, n)
    counts = [int(c) for c in counts.tolist()]
    bmax = max(counts)
    send = torch.zeros(bmax, dtype=frames.dtype, device=frames.device)
    send[: frames.numel()] = frames
    out = torch.empty(world * bmax, dtype=frames.dtype, device=frames.device)
    dist.all_gather_into_tensor(out, send)
    return torch.cat([out[r * bmax: r * bmax + counts[r]] for r in range(world)])


def gather_pcm(pcm, lengths):
    """pcm: [B, cap] fp32 (or int16, see to_pcm16) on this rank's device (rows valid up to lengths[b]); lengths: [B] int64 (same device).
    Returns (gathered [sum of B over ranks, smax], all_lengths) on every rank, rank blocks in rank order; world == 1 is a
    no-op view. B may differ between ranks (shard_range hands out blocks that differ by one): the row counts travel first
    and short blocks are padded to the largest for the fixed-size collective, then the padding rows are dropped."""
    B = pcm.shape[0]
    if not dist.is_available() or not dist.is_initialized():
        smax = int(lengths.max().item()) if B else 0
        return pcm[:, :smax], lengths
    world = dist.get_world_size()
    # one small control message: [B, longest length] of this rank -> every rank (a rank whose shard is empty — fewer utterances
    # than ranks — sends [0, 0] and an all-padding block)
    own_max = lengths.max().to(torch.int64) if B else torch.zeros((), dtype=torch.int64, device=lengths.device)
    meta = torch.stack([torch.tensor(B, dtype=torch.int64, device=lengths.device), own_max])
    metas = torch.empty(2 * world, dtype=torch.int64, device=lengths.device)
    dist.all_gather_into_tensor(metas, meta)
    metas = metas.view(world, 2).tolist()
    counts = [int(m[0]) for m in metas]
    smax = max(int(m[1]) for m in metas)
    bmax = max(counts)
    if pcm.shape[1] < smax:
        raise ValueError("PCM buffer narrower than the longest utterance of another rank: all ranks must use the same capacity")
    len_send = lengths.contiguous()
    send = pcm[:, :smax].contiguous()
    if B < bmax:  # pad this rank's block to the common row count
        len_send = torch.cat([len_send, torch.zeros(bmax - B, dtype=lengths.dtype, device=lengths.device)])
        send = torch.cat([send, torch.zeros((bmax - B, smax), dtype=pcm.dtype, device=pcm.device)])
    all_len = torch.empty(world * bmax, dtype=lengths.dtype, device=lengths.device)
    dist.all_gather_into_tensor(all_len, len_send)
    out = torch.empty((world * bmax, smax), dtype=pcm.dtype, device=pcm.device)
    if pcm.dtype == torch.float32:
        dist.all_gather_into_tensor(out, send)
    else:
        # an all-gather only copies: int16 (which neither RCCL nor gloo has as an element type) travels as bytes
        dist.all_gather_into_tensor(out.view(torch.uint8), send.view(torch.uint8))
    if all(c == bmax for c in counts):
        return out, all_len
    keep = torch.cat([torch.arange(r * bmax, r * bmax + counts[r], device=out.device) for r in range(world)])
    return out.index_select(0, keep), all_len.index_select(0, keep.to(all_len.device))


def gather_index_map(shards):
    """Row r of a gathered block holds global utterance gather_index_map(shards)[r]: rank blocks arrive in rank order, so with
    `balanced_shards` (or any non-contiguous assignment) the gathered rows are NOT in global utterance order."""
    return [i for s in shards for i in s]


def restore_order(gathered, lengths, shards):
    """Reorders the rows of a gather over `shards` (list of per-rank index lists) into global utterance order."""
    idx = torch.as_tensor(gather_index_map(shards), dtype=torch.int64, device=gathered.device)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(idx.numel(), device=idx.device)
    return gathered.index_select(0, inv), lengths.index_select(0, inv.to(lengths.device))


class PcmExchange:
    """The path's one exchange as a PIPELINE with no host round trip and no allocation in the steady state.

    `gather_pcm` above is the simple synchronous form: a meta all-gather + `.tolist()` (two host syncs) and three allocations per
    call, all inside the step. Here everything that does not change is agreed ONCE (row counts per rank, row capacity), every
    buffer is allocated once, and step i's traffic overlaps step i + 1's compute:

      submit(pcm_i, lengths_i)   (1) queues the fixed-size all-gather of the int64 lengths of step i and their copy to pinned
                                     host memory on the exchange stream — nothing waits for it;
                                 (2) reads the lengths of step i-1 (that copy finished a whole step ago), which gives the exact
                                     common width smax_{i-1}, and queues the all-gather of PCM_{i-1}[:, :smax] on the exchange stream;
      flush()                    drains the pipeline (end of a run, or when the caller needs the last block).

    The PCM of step i is therefore read by the collective while step i + 1 runs: the caller must not overwrite it before step
    i + 2 begins (rotate THREE output buffers: `slot(i) = i % 3`; submit(i + 2) returns only after the copy of PCM i out of its buffer
    has completed, so the rotation is safe however far the exchange stream lags). All device work of the exchange — the copies into the
    send buffer, the collectives, the removal of padding rows — runs on the exchange stream, ordered behind the caller's stream at every
    submit.
    Results arrive through `on_block(step, gathered [rows, smax], lengths [rows])` (views into the exchange's own buffers, valid
    until the next block is produced) or `last`. Works on CPU tensors with gloo (synchronous there) and on the GPU with
    nccl == RCCL; int16 PCM travels as bytes."""

    def __init__(self, rows, cap, dtype=torch.float32, device="cpu", on_block=None, row_capacity=0):
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.dist_on else 1
        self.rank = dist.get_rank() if self.dist_on else 0
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.cap, self.dtype, self.on_block = int(cap), dtype, on_block
        # agreed once: rows per rank (shards may differ by one, or be empty)
        if self.dist_on:
            mine = torch.tensor([rows], dtype=torch.int64, device=self.device)
            allc = torch.empty(self.world, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, mine)
            self.counts = [int(x) for x in allc.tolist()]
        else:
            self.counts = [int(rows)]
        self.rows = int(rows)
        # every rank's block is padded to bmax rows (row_capacity: a caller whose shard sizes change from run to run fixes it up front)
        self.bmax = max(max(self.counts), 1, int(row_capacity))
        n = self.world * self.bmax
        self.keep = self.keep_cpu = self.kept = None
        if any(c != self.bmax for c in self.counts):
            self.keep_cpu = torch.cat([torch.arange(r * self.bmax, r * self.bmax + self.counts[r]) for r in range(self.world)])
            self.keep = self.keep_cpu.to(self.device)
            self.kept = torch.zeros(int(self.keep_cpu.numel()) * self.cap, dtype=dtype, device=self.device)  # the gathered block without the padding rows
        pin = dict(pin_memory=True) if self.cuda else {}
        self.len_send = [torch.zeros(self.bmax, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.len_all = [torch.zeros(n, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.len_host = [torch.zeros(n, dtype=torch.int64, **pin) for _ in range(2)]
        self.send = torch.zeros(self.bmax * self.cap, dtype=dtype, device=self.device)  # (rows beyond `rows` are the padding block: never read back, see `keep`)
        self.out = torch.zeros(n * self.cap, dtype=dtype, device=self.device)
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.len_ev = [torch.cuda.Event() if self.cuda else None for _ in range(2)]
        self.out_ev = torch.cuda.Event() if self.cuda else None
        self.copy_ev = torch.cuda.Event() if self.cuda else None  # behind the copy of a step's PCM into `send`: the caller's buffer is free again
        self.copy_pending = False
        self.pending = None  # (step, pcm, slot): lengths queued, PCM not yet
        self.step = 0
        self.last = None
        self.bytes_moved = 0

    def _on_side(self):
        return torch.cuda.stream(self.side) if self.cuda else _NullCtx()

    def submit(self, pcm, lengths):
        """pcm [rows, >= cap used] on self.device (fp32 or int16), rows valid up to lengths[b]; lengths int64 [rows] on the same
        device, both produced on (or synchronised with) the CALLER's current stream: the exchange stream is ordered behind it here.
        Returns without waiting for any collective; pcm must stay untouched until the SECOND next submit has returned / the next flush
        (rotate three buffers): submit(i) returns only when the copy of PCM i - 2 out of its buffer has completed."""
        assert pcm.shape[0] == self.rows and pcm.dtype == self.dtype and pcm.shape[1] <= self.cap
        slot = self.step & 1
        # (the caller's stream must be looked up OUTSIDE the side-stream context: inside it, the current stream IS the side stream)
        producer = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.cuda and self.copy_pending:
            # the previous submit queued the copy of PCM step-2 into `send`; the caller is about to reuse that buffer for step + 1
            self.copy_ev.synchronize()
            self.copy_pending = False
        with self._on_side():
            if self.cuda:
                self.side.wait_stream(producer)
            self.len_send[slot][: self.rows].copy_(lengths)
            if self.dist_on:
                dist.all_gather_into_tensor(self.len_all[slot], self.len_send[slot])
            else:
                self.len_all[slot].copy_(self.len_send[slot])
            self.len_host[slot].copy_(self.len_all[slot], non_blocking=True)
            if self.cuda:
                self.len_ev[slot].record(self.side)
        prev, self.pending = self.pending, (self.step, pcm, slot)
        self.step += 1
        if prev is not None:
            self._gather(*prev, producer=producer)

    def _gather(self, step, pcm, slot, producer=None):
        if self.cuda:
            self.len_ev[slot].synchronize()  # (recorded a whole step ago: no wait in the steady state)
        lens_host = self.len_host[slot]
        smax = int(lens_host.max())
        if smax > pcm.shape[1]:
            raise ValueError("PCM buffer narrower than the longest utterance of another rank: all ranks must use the same capacity")
        n = self.world * self.bmax
        if self.cuda and producer is None:
            producer = torch.cuda.current_stream(self.device)
        with self._on_side():
            if self.cuda:
                self.side.wait_stream(producer)  # pcm was written on the caller's stream (e.g. the int16 conversion bench.py launches there)
            send = self.send[: self.bmax * smax].view(self.bmax, smax)
            send[: self.rows].copy_(pcm[:, :smax])
            if self.cuda:
                self.copy_ev.record(self.side)
                self.copy_pending = True
            out = self.out[: n * smax].view(n, smax)
            if self.dist_on:
                if self.dtype == torch.float32:
                    dist.all_gather_into_tensor(out, send)
                else:  # an all-gather only copies: int16 (not an element type of RCCL / gloo) travels as bytes
                    dist.all_gather_into_tensor(out.view(torch.uint8), send.view(torch.uint8))
            else:
                out.copy_(send)
            if self.keep is not None:
                # drop the padding rows ON THE EXCHANGE STREAM, behind the all-gather that fills `out`, into a buffer of the exchange
                # (no allocation, and nothing on another stream reads `out` before it is complete)
                kept = self.kept[: self.keep_cpu.numel() * smax].view(-1, smax)
                torch.index_select(out, 0, self.keep, out=kept)
                out = kept
            if self.cuda:
                self.out_ev.record(self.side)
        self.bytes_moved += n * smax * out.element_size()
        lens = lens_host.clone()
        if self.keep is not None:
            lens = lens.index_select(0, self.keep_cpu)
        self.last = (step, out, lens)
        if self.on_block is not None:
            if self.cuda:
                self.out_ev.synchronize()
            self.on_block(step, out, lens)

    def flush(self):
        """Queue the PCM gather of the last submitted step and wait for everything in flight."""
        prev, self.pending = self.pending, None
        if prev is not None:
            self._gather(*prev)
        if self.cuda:
            self.side.synchronize()
            self.copy_pending = False
        return self.last


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
