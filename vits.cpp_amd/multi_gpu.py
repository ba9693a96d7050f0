"""Multi-GPU plumbing: one process per GPU, utterances sharded across ranks, ONE exchange — the all-gather of the PCM.

The reference processes exactly one utterance per call (src/vits.cpp:184,303), so utterances are independent: rank r
synthesises utterances [r*B, (r+1)*B) with replicated weights and no data-path collective. The only exchange is the final
gather of fp32 PCM (ragged: per-utterance lengths travel first), over torch.distributed (backend "nccl" == RCCL on ROCm,
xGMI between the 8 GPUs of a node; "gloo" on CPU for the tests). Payload at the benchmark shape: 64 x ~58k x 4 B ~ 15 MB
per rank — negligible next to the >100 ms of compute per step, so a plain all_gather_into_tensor is used.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block of utterances owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


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


def gather_pcm(pcm, lengths):
    """pcm: [B, cap] fp32 (or int16, see to_pcm16) on this rank's device (rows valid up to lengths[b]); lengths: [B] int64 (same device).
    Returns (gathered [world*B, smax] fp32, all_lengths [world*B] int64) on every rank; world == 1 is a no-op view."""
    if not dist.is_available() or not dist.is_initialized():
        smax = int(lengths.max().item())
        return pcm[:, :smax], lengths
    world = dist.get_world_size()
    B = pcm.shape[0]
    smax_t = lengths.max().clone()
    dist.all_reduce(smax_t, op=dist.ReduceOp.MAX)
    smax = int(smax_t.item())
    if pcm.shape[1] < smax:
        raise ValueError("PCM buffer narrower than the longest utterance of another rank: all ranks must use the same capacity")
    all_len = torch.empty(world * B, dtype=lengths.dtype, device=lengths.device)
    dist.all_gather_into_tensor(all_len, lengths.contiguous())
    send = pcm[:, :smax].contiguous()
    out = torch.empty((world * B, smax), dtype=pcm.dtype, device=pcm.device)
    if pcm.dtype == torch.float32:
        dist.all_gather_into_tensor(out, send)
    else:
        # an all-gather only copies: int16 (which neither RCCL nor gloo has as an element type) travels as bytes
        dist.all_gather_into_tensor(out.view(torch.uint8), send.view(torch.uint8))
    return out, all_len
