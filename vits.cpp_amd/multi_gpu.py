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
    if not dist.is_available() or not dist.is_initialized():
        smax = int(lengths.max().item())
        return pcm[:, :smax], lengths
    world = dist.get_world_size()
    B = pcm.shape[0]
    # one small control message: [B, longest length] of this rank -> every rank
    meta = torch.stack([torch.tensor(B, dtype=torch.int64, device=lengths.device), lengths.max().to(torch.int64)])
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
