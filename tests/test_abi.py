"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/vits.h declares, and fails loudly
(no CPU fallback) when there is no GPU. No compute calls here."""
import os
import re

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
    return strstr(vits_last_error(), "failed to open file") ? 0 : 3;
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
