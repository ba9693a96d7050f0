"""Test-side numpy parser of the reference's model format — a third implementation, independent of
the product reader (vits.cpp_amd/csrc/model_file.cpp) and of the oracle reader (oracle/vits_oracle.cpp)."""
import struct

import numpy as np


def parse_model_file(data):
    off = 0

    def u32():
        nonlocal off
        (v,) = struct.unpack_from("<I", data, off)
        off += 4
        return v

    def s():
        nonlocal off
        n = u32()
        v = data[off:off + n].decode("utf-8")
        off += n
        return v

    vocab = {}
    for _ in range(u32()):
        k = s()
        vocab[k] = u32()
    add_blank, normalize = u32(), u32()
    pad, unk = s(), s()
    cfg = {}
    for _ in range(u32()):
        k = s()
        cfg[k] = s()
    tensors = {}
    for _ in range(u32()):
        name = s()
        dt, rank = u32(), u32()
        ne = [u32() for _ in range(rank)]
        nb = u32()
        arr = np.frombuffer(data, dtype=np.float32 if dt == 0 else np.float16, count=nb // (4 if dt == 0 else 2), offset=off)
        off += nb
        tensors[name] = (arr.reshape(ne[::-1]).copy(), dt)
    assert off == len(data)
    return dict(vocab=vocab, add_blank=add_blank, normalize=normalize, pad=pad, unk=unk, config=cfg, tensors=tensors)
