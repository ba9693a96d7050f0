"""The product's host code under AddressSanitizer + UBSan and a mutation fuzzer: the code that eats untrusted bytes in
vits_model_load_from_bytes — container parser, hyper-parameter reader, per-tensor shape validation against them, weight packing, tokenizer, writer, the ABI
layer — is rebuilt with -fsanitize=address,undefined (`make -C vits.cpp_amd/csrc asan`: the host translation units sanitized, the kernel objects as shipped) and
driven by tests/host_fuzz.cpp through the C ABI with 20,000 seeded mutations of the exporter-written fixture: truncations, byte flips, extreme values in every
length / count / type / rank / dimension / byte-length word. The reference's reader trusts the file (/root/reference/src/vits_model_data.cpp:29-97,
/root/reference/src/vits_tokenizer.cpp:22-78); here every malformed file must come back as an error with a message, with no sanitizer report."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vits.cpp_amd", "csrc")


def test_host_code_is_clean_under_asan_ubsan_and_a_mutation_fuzzer():
    jobs = str(min(8, os.cpu_count() or 1))
    mk = subprocess.run(["make", "-s", "-j", jobs, "-C", CSRC, "asan"], capture_output=True, text=True, timeout=1500)
    assert mk.returncode == 0, mk.stderr[-4000:]
    exe = os.path.join(CSRC, "_asan", "host_fuzz")
    n = int(os.environ.get("VITS_FUZZ_MUTATIONS", "20000"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "tiny_hf_export.ggml"), str(n), "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    m = re.search(r"host_fuzz ok: (\d+) mutations .* (\d+) validated, (\d+) rejected, (\d+) reserialized, (\d+) tokenized", r.stdout)
    assert m, r.stdout
    total, validated, rejected, reser, tok = map(int, m.groups())
    assert total == n and validated + rejected == n
    # the corpus exercises both sides: a good share of the mutants still loads, most are refused
    assert validated > n // 50 and rejected > n // 2 and reser >= validated
