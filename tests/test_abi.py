"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/vits.h declares, and fails loudly
(no CPU fallback) when there is no GPU. No compute calls here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vits.h")).read()
    return sorted(set(re.findall(r"VITS_API\s+[\w\s\*]+?\b(vits_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/vits.h but not exported"
    assert sorted(pkg.EXPORTED_SYMBOLS) == declared


def test_reference_five_symbols_present(pkg):
    # /root/reference/src/include/vits.h:94-102
    for name in ["vits_model_load_from_bytes", "vits_model_load_from_file", "vits_free_model", "vits_free_result", "vits_model_process"]:
        assert hasattr(pkg.lib(), name)


def test_no_gpu_means_loud_failure_not_a_cpu_fallback(pkg, tiny_bytes):
    try:
        pkg.device_info()
        has_gpu = True
    except pkg.VitsError:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.VitsError, match="no HIP device"):
        pkg.Model(tiny_bytes)


def test_bad_inputs_return_null_with_message(pkg):
    L = pkg.lib()
    assert not L.vits_model_load_from_bytes(b"\x01\x00\x00\x00", 4)
    assert "truncated" in pkg.last_error()
    assert not L.vits_model_load_from_file(b"/nonexistent/model.ggml")
    assert "failed to open file" in pkg.last_error()  # message of the reference (vits_model_data.cpp:102)


def test_arith_and_scope_setters_reject_bad_arguments_without_a_model(pkg):
    """The mode / arithmetic / scope setters of include/vits.h never dereference a NULL handle and never accept an unknown value: -1 and a
    message, nothing crosses the boundary (the reference's API has no such setters: /root/reference/src/include/vits.h:87-102)."""
    L = pkg.lib()
    assert L.vits_model_set_arith_scope(None, pkg.SCOPE_ALL_CONVS) == -1 and "scope" in pkg.last_error()
    assert L.vits_model_get_arith_scope(None) == -1
    assert L.vits_model_set_arith(None, pkg.ARITH_F16) == -1
    assert L.vits_model_get_arith(None) == -1
    assert L.vits_model_set_mode(None, pkg.MODE_HF) == -1


def test_product_does_not_link_or_load_the_oracle(pkg):
    import subprocess
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    src_dir = os.path.join(ROOT, "vits.cpp_amd")
    for dirpath, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".py")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vits_oracle" not in text and "oracle_lib" not in text, f


def test_pcm16_and_wav_writer_match_the_reference_driver(pkg, tmp_path):
    """/root/reference/test/main.cpp:23-63: clamp to [-1,1], * 32767, truncating cast; canonical 44-byte header, 16 kHz mono."""
    import struct

    import numpy as np
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1.7, -3.0, 0.99999, -1e-9, 3.0517578e-05], np.float32)
    want = (np.clip(x, -1.0, 1.0) * np.float32(32767)).astype(np.int16)  # C cast truncates toward zero, like numpy astype
    np.testing.assert_array_equal(pkg.pcm16(x), want)
    path = tmp_path / "o.wav"
    pkg.write_wav16(str(path), x, 16000)
    raw = path.read_bytes()
    assert len(raw) == 44 + 2 * x.size
    riff, size, wave, fmt, fsz, afmt, ch, sr, br, align, bits, data, dbytes = struct.unpack("<4sI4s4sIHHIIHH4sI", raw[:44])
    assert (riff, wave, fmt, data) == (b"RIFF", b"WAVE", b"fmt ", b"data")
    assert (fsz, afmt, ch, sr, br, align, bits, dbytes) == (16, 1, 1, 16000, 32000, 2, 16, 2 * x.size)
    assert size == 4 + (8 + 16) + (8 + dbytes)
    np.testing.assert_array_equal(np.frombuffer(raw[44:], np.int16), want)


def test_header_is_plain_c_and_links_from_a_c_caller(pkg, tmp_path):
    """include/vits.h must compile as C99 (the reference's consumers are C / Swift FFI, Makefile:21-24) and a C program
    using the reference's five entry points must link against libvits_hip.so and get NULL + a message — not a crash — for a
    missing file (the reference throws std::runtime_error across the boundary there, src/vits_model_data.cpp:102)."""
    import subprocess
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "vits.h"
static int sink(void* user, int32_t utt, size_t offset, const float* pcm, size_t n) { (void)user; (void)utt; (void)offset; (void)pcm; (void)n; return 0; }
int main(void) {
    vits_process_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.on_chunk = sink;
    vits_model* m = vits_model_load_from_file("/nonexistent/model.ggml");
    if (m) { vits_result r = vits_model_process(m, "hello"); vits_free_result(r); vits_free_model(m); return 2; }
    printf("%s\n", vits_last_error());
    if (!strstr(vits_last_error(), "failed to open file")) return 3;
    /* the multi-GPU gather from C: bad arguments are refused with a message; a one-rank object needs a device (none here: NULL, no crash) */
    {
        vits_gather_result all;
        char id[VITS_GATHER_ID_BYTES];
        vits_gather_ctx* g;
        memset(id, 0, sizeof id);
        if (vits_pcm_gather_init(id, sizeof id, 2, 2, 4, 100, 4) != NULL) return 4;  /* rank >= world */
        if (!strstr(vits_last_error(), "rank")) return 5;
        if (vits_pcm_gather(NULL, NULL, 0, NULL, NULL, &all) != -1) return 6;
        g = vits_pcm_gather_init(NULL, 0, 0, 1, 4, 100, 4);
        if (g) vits_pcm_gather_destroy(g);  /* (a GPU box: fine) */
        else printf("%s\n", vits_last_error());
        vits_pcm_gather_destroy(NULL);
    }
    return 0;
}
''')
    exe = tmp_path / "caller"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(pkg.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lvits_hip",
                         "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)


def test_busy_guard_admits_one_caller(tmp_path):
    """csrc/busy_guard.h (the flag the C ABI takes around every entry point of a model handle: a second caller gets "model busy"
    instead of racing on the arenas; reference contract: src/include/vits.h:22-30) as a CPU unit test: eight threads contend for one
    flag — never two inside, refusals do happen, a nested attempt from the owner is refused, the flag is free afterwards."""
    import subprocess
    src = tmp_path / "busy.cpp"
    src.write_text(r'''
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#include "busy_guard.h"
int main() {
    std::atomic<bool> flag{false};
    std::atomic<int> inside{0}, worst{0}, entered{0}, refused{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([&] {
            for (int i = 0; i < 20000; ++i) {
                vits::BusyGuard g(&flag);
                if (!g.entered()) { ++refused; continue; }
                const int now = ++inside;
                int w = worst.load();
                while (now > w && !worst.compare_exchange_weak(w, now)) {}
                vits::BusyGuard nested(&flag);  // a callback re-entering its own handle
                if (nested.entered()) worst = 99;
                ++entered;
                --inside;
            }
        });
    for (auto& t : th) t.join();
    vits::BusyGuard none(nullptr);
    std::printf("%d %d %d %d %d\n", worst.load(), entered.load(), refused.load(), (int)flag.load(), (int)none.entered());
    return 0;
}
''')
    exe = tmp_path / "busy"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Werror", "-I", os.path.join(ROOT, "vits.cpp_amd", "csrc"), str(src), "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    worst, entered, refused, flag, none = map(int, subprocess.run([str(exe)], capture_output=True, text=True).stdout.split())
    assert worst == 1 and entered > 0 and entered + refused == 8 * 20000 and flag == 0 and none == 0
    # and the ABI uses it at every entry point that touches an engine
    abi = open(os.path.join(ROOT, "vits.cpp_amd", "csrc", "abi.cpp")).read()
    assert abi.count("VITS_ENTER(model") >= 12


# ---- model files whose tensor shapes disagree with the hyper-parameters must be rejected at load ------------------------
def _tensor_headers(data):
    """(name, header offset of `rank`, rank, ne list, nbytes) for every tensor record of a model file."""
    import struct
    off = 0

    def u32():
        nonlocal off
        (v,) = struct.unpack_from("<I", data, off)
        off += 4
        return v

    def skip_str():
        nonlocal off
        n = u32()
        s = data[off:off + n]
        off += n
        return s

    for _ in range(u32()):
        skip_str()
        u32()
    u32(), u32()
    skip_str(), skip_str()
    for _ in range(u32()):
        skip_str(), skip_str()
    out = []
    for _ in range(u32()):
        name = skip_str().decode()
        u32()  # dtype
        rank_off = off
        rank = u32()
        ne = [u32() for _ in range(rank)]
        nb = u32()
        out.append((name, rank_off, rank, ne, nb))
        off += nb
    return out


def _reshape_tensor(data, name, new_ne):
    """Same payload, different declared shape (same rank so that the record keeps its size)."""
    import struct
    for n, rank_off, rank, ne, nb in _tensor_headers(data):
        if n == name:
            assert len(new_ne) == rank, (ne, new_ne)
            b = bytearray(data)
            struct.pack_into("<" + "I" * rank, b, rank_off + 4, *new_ne)
            return bytes(b), ne
    raise KeyError(name)


@pytest.mark.parametrize("name,reshape", [
    ("text_encoder.encoder.layers.0.attention.q_proj.weight", lambda ne: [ne[0] // 2, ne[1] * 2]),
    ("text_encoder.embed_tokens.weight", lambda ne: [ne[1], ne[0]]),
    ("text_encoder.encoder.layers.1.attention.emb_rel_k", lambda ne: [ne[1], ne[0], ne[2]]),
    ("text_encoder.encoder.layers.0.feed_forward.conv_1.weight", lambda ne: [ne[1], ne[0], ne[2]]),
    ("duration_predictor.conv_dds.convs_dilated.0.weight", lambda ne: [ne[2], ne[1], ne[0]]),
    ("duration_predictor.flows.2.conv_proj.weight", lambda ne: [ne[0], ne[2], ne[1]]),
    ("flow.flows.0.wavenet.in_layers.0.weight", lambda ne: [ne[0], ne[2], ne[1]]),
    ("decoder.upsampler.0.weight", lambda ne: [ne[0] * 2, ne[1] // 2, ne[2]]),
    ("decoder.resblocks.0.convs1.0.weight", lambda ne: [ne[1], ne[0], ne[2]]),
    ("decoder.conv_post.weight", lambda ne: [ne[1], ne[0], ne[2]]),
    ("decoder.conv_pre.bias", lambda ne: [ne[0] // 2]),
])
def test_tensor_shapes_are_checked_against_the_hyper_parameters(pkg, tiny_bytes, name, reshape):
    """shapes from the file were trusted (host over-read in the q/k/v memcpy, device out-of-bounds reads for
    embeddings / depthwise / rel_k). Every tensor is now checked at load; vits_model_file_validate runs the same checks
    without a device. The payload size still matches the declared shape here, so only the shape check can catch it."""
    import struct
    pkg.validate(tiny_bytes)  # the unmodified file passes
    if name.endswith(".bias"):
        # a bias with half the elements: shrink the record (rank 1)
        for n, rank_off, rank, ne, nb in _tensor_headers(tiny_bytes):
            if n == name:
                new_n = reshape(ne)[0]
                esz = nb // ne[0]
                b = bytearray(tiny_bytes[:rank_off + 4]) + struct.pack("<II", new_n, new_n * esz) + tiny_bytes[rank_off + 12:rank_off + 12 + new_n * esz] + \
                    tiny_bytes[rank_off + 12 + nb:]
                bad = bytes(b)
                break
    else:
        ne0 = next(h[3] for h in _tensor_headers(tiny_bytes) if h[0] == name)
        bad, _ = _reshape_tensor(tiny_bytes, name, reshape(ne0))
    with pytest.raises(pkg.VitsError) as ei:
        pkg.validate(bad)
    assert name.rsplit(".", 1)[0] in str(ei.value), str(ei.value)  # the message names the offending tensor
    assert not pkg.lib().vits_model_load_from_bytes(bad, len(bad))  # the loader refuses it too (with or without a GPU)


def test_tensor_dimension_product_overflow_is_rejected(pkg, tiny_bytes):
    """Four u32 dimensions can wrap a 64-bit product back to the payload size; the parser counts with an overflow guard."""
    import struct
    name, rank_off, rank, ne, nb = _tensor_headers(tiny_bytes)[0]
    # rebuild the record with rank 4 and dims whose 64-bit product wraps to the true element count
    esz = nb // int(__import__("numpy").prod(ne))
    count = nb // esz
    dims = [1 << 31, 1 << 31, 4, count]  # 2^64 * count == count (mod 2^64) for the product of the first three = 2^64
    rec = struct.pack("<I", 4) + struct.pack("<IIII", *dims) + struct.pack("<I", nb)
    bad = tiny_bytes[:rank_off] + rec + tiny_bytes[rank_off + 4 + 4 * rank + 4:]
    with pytest.raises(pkg.VitsError, match="overflows|byte length"):
        pkg.validate(bad)


def _patch_model_file(data, add_blank=None, extra_cfg=()):
    """Rewrite two header fields of a model file: the add_blank word and the config block (extra key / value pairs appended)."""
    import struct
    off = 0

    def u32():
        nonlocal off
        (v,) = struct.unpack_from("<I", data, off)
        off += 4
        return v

    def skip_s():
        nonlocal off
        n = u32()
        off += n

    for _ in range(u32()):
        skip_s()
        u32()
    ab_off = off
    u32(), u32()
    skip_s(), skip_s()
    ncfg_off = off
    ncfg = u32()
    for _ in range(ncfg):
        skip_s(), skip_s()
    cfg_end = off
    out = bytearray(data[:cfg_end])
    if add_blank is not None:
        struct.pack_into("<I", out, ab_off, add_blank)
    struct.pack_into("<I", out, ncfg_off, ncfg + len(extra_cfg))
    for k, v in extra_cfg:
        for t in (k.encode(), v.encode()):
            out += struct.pack("<I", len(t)) + t
    return bytes(out) + data[cfg_end:]


def test_text_entry_points_honour_phonetic_and_add_blank(pkg, tiny_hf_bytes):
    """The two model-file flags of the tokenizer (VERDICT r5 missing 4 / weak 12). `phonetic = 1` (vits_model_data.cpp:92-94 -> set_phonetic,
    vits_tokenizer.cpp:176-178: the reference asserts out without VITS_ESPEAK): the file still validates and the text entry point refuses with a message that
    says to pass ids; `phonetic = 0` changes nothing. `add_blank = 0` (Q11, vits_tokenizer.cpp:200-208: tokens_final is only filled under add_blank): no ids."""
    base = pkg.file_tokenize(tiny_hf_bytes, "hello world")
    assert base.size == 2 * 11 + 1 and (base[0::2] == base[0]).all()  # eleven known symbols (the space is one), blanks interspersed
    ph = _patch_model_file(tiny_hf_bytes, extra_cfg=[("phonetic", "1")])
    pkg.validate(ph)  # (raises on failure: a phonetic model LOADS; the reference without VITS_ESPEAK would assert out)
    with pytest.raises(pkg.VitsError) as e:
        pkg.file_tokenize(ph, "hello world")
    assert "espeak" in str(e.value) and "pass ids" in str(e.value)
    assert np.array_equal(pkg.file_tokenize(_patch_model_file(tiny_hf_bytes, extra_cfg=[("phonetic", "0")]), "hello world"), base)
    nb = _patch_model_file(tiny_hf_bytes, add_blank=0)
    assert pkg.file_tokenize(nb, "hello world").size == 0
    assert pkg.reserialize(ph) == ph and pkg.reserialize(nb) == nb  # both are well-formed files
