"""Every test the documents cite must exist (commit 9ba1a14 silently deleted two tests that DESIGN.md and
include/vits_exact_math.h still named as the independent evidence behind a parity claim).

A citation is `tests/test_x.py::test_name`, `test_x.py::test_name` or a bare `test_name` in back-ticks / prose; a name that ends in `_`, `…` or
`*` is a prefix citation (DESIGN.md abbreviates long names) and must match at least one collected test. File citations (`tests/test_x.py`) must
name files that exist. The reference's own `test/test_ggml_utils.cpp` is not ours and is skipped."""
import ast
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "docs/*.md", "include/*.h", "oracle/*.h", "oracle/*.cpp",
        "vits.cpp_amd/csrc/*.h", "vits.cpp_amd/csrc/*.cpp", "vits.cpp_amd/csrc/*.hip", "vits.cpp_amd/*.py", "bench.py", "__graft_entry__.py"]
NOT_OURS = {"test_ggml_utils", "test_vits", "test_tokenizer", "test_depthwise", "test_transpose", "test_remove_weight_norm"}  # /root/reference files


def collected():
    """name -> set of files, from the AST of tests/test_*.py (functions and methods; what pytest collects by its default rules)."""
    names, files = {}, set()
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py"))):
        files.add(os.path.basename(path)[:-3])
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and node.name.startswith("test_"):
                names.setdefault(node.name, set()).add(os.path.basename(path)[:-3])
    return names, files


def citations():
    out = []
    for pat in DOCS:
        for path in sorted(glob.glob(os.path.join(ROOT, pat))):
            text = open(path, errors="replace").read()
            for m in re.finditer(r"(?:(test_[A-Za-z0-9_]+)\.py::)?(test_[A-Za-z0-9_]+)([…*]|\.\.\.)?", text):
                out.append((os.path.relpath(path, ROOT), text.count("\n", 0, m.start()) + 1, m.group(1), m.group(2), bool(m.group(3))))
    return out


def test_every_cited_test_is_collected():
    names, files = collected()
    missing = []
    for doc, line, file_, name, ellipsis in citations():
        if name in NOT_OURS or (file_ and file_ in NOT_OURS):
            continue
        if name in files and not file_:      # a file citation such as tests/test_oracle.py
            continue
        prefix = ellipsis or name.endswith("_")
        if not file_ and any(f.startswith(name) for f in files) and (prefix or name == "test_gpu"):  # tests/test_gpu_*.py, "the test_gpu files"
            continue
        hits = [n for n in names if (n.startswith(name) if prefix else n == name)]
        if file_:
            if file_ not in files:
                missing.append("%s:%d cites a test file that does not exist: %s.py" % (doc, line, file_))
                continue
            hits = [n for n in hits if file_ in names[n]]
        if not hits:
            missing.append("%s:%d cites %s%s%s, which pytest does not collect" % (doc, line, (file_ + ".py::") if file_ else "", name, "…" if prefix else ""))
    assert not missing, "\n".join(missing)


def test_the_citation_scan_sees_what_it_should():
    """The scan itself: the two tests whose deletion prompted this file are cited, and they are found."""
    cited = {(n, e) for _, _, _, n, e in citations()}
    assert any(n.startswith("test_exact_math_header_") for n, _ in cited)
    assert any(n.startswith("test_oracle_exact_order_table_mode_") for n, _ in cited)
    names, files = collected()
    assert "test_exact_math_header_conversions_and_polynomials" in names
    assert "test_oracle_exact_order_table_mode_agrees_with_its_independent_loops" in names
    assert "test_oracle" in files and len(names) > 120
