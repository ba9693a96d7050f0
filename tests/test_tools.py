"""Host-side tooling that bench.py's roofline block depends on (no GPU): the rocprofv3 kernel name -> bench key map of
tools/pmc_common.py must know every matrix-core kernel family of the library. A family it does not recognise turns into `null`
traffic / MFMA-busy fields in the bench line with no warning."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

MFMA_FAMILIES = ("conv_mfma_kernel", "conv_group_kernel", "conv16_kernel", "rbpair16_kernel", "rbpair32_kernel", "rbblock16_kernel", "wavenet16_kernel",
                 "wavenet32_kernel", "flow_couple16_kernel", "convt16_kernel", "convt16_lines_kernel")


def test_every_matrix_core_kernel_of_the_committed_traces_has_a_bench_key():
    from pmc_common import bench_key
    names = set()
    # (traces of the current kernel templates: the last set of round 3 and everything since; older rounds had other template lists)
    for path in glob.glob(os.path.join(ROOT, "profiles", "round3_v13*kernel_stats.csv")) + glob.glob(os.path.join(ROOT, "profiles", "round[4-9]*kernel_stats.csv")):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                names.add(row["Name"])
    assert len(names) > 50
    seen = set()
    for n in sorted(names):
        m = re.search(r"vits::(\w+)<", n)
        if not m or m.group(1) not in MFMA_FAMILIES:
            continue
        seen.add(m.group(1))
        key = bench_key(n)
        assert key is not None, n
        assert re.fullmatch(r"k[\dG]+\|d-?\d+\|[A-Za-z][\w.]*\|e\d\w*", key), (n, key)
    # (wavenet16_kernel only runs with VITS_NO_FLOW_FUSE=1 since the coupling layers became one kernel: not in the default traces)
    assert seen >= set(MFMA_FAMILIES) - {"wavenet16_kernel"}, set(MFMA_FAMILIES) - seen
    assert bench_key("void vits::wavenet16_kernel<192, 5, false, 1>(vits::WaveNet16Params)") == "k5|d1|W192|e1"
    # the names the round-3 regexes missed, spelled out
    assert bench_key("void vits::flow_couple16_kernel<false, 2>(vits::FlowCouple16Params)") == "k5|d1|C192|e1"
    assert bench_key("void vits::rbblock16_kernel<11, 32, 4, 3, 1, 1, 3, 5, false>(vits::RbBlockParams)") == "k11|d135|B32|e0g"
    assert bench_key("void vits::convt16_kernel<4, 1, 16, false>(vits::ConvT16Params)") == "k2|d-1|S4.1.16|e2g"
    assert bench_key("void vits::convt16_lines_kernel<64, false>(vits::ConvT16Params)") == "k2|d-1|SL64|e2g"
    assert bench_key("void vits::rbblock32_kernel<32, 2>(vits::RbBlock32Params)") == "k3|d135|b32|e0"


def test_the_engine_labels_use_the_same_tile_tags():
    """The profiler label of the streaming upsamplers comes from convt16_stream_tag (same place that picks the instantiation)."""
    src = open(os.path.join(ROOT, "vits.cpp_amd", "csrc", "engine_vocoder.cpp")).read()
    assert "convt16_stream_tag(U.up, tag, sizeof(tag))" in src
    ct = open(os.path.join(ROOT, "vits.cpp_amd", "csrc", "convt16.hip")).read()
    assert '"SL%d"' in ct and '"S%d.%d.%d"' in ct


def test_default_schedule_reducer_on_a_committed_trace(tmp_path):
    """tools/default_schedule.py (the library-default schedule's rocprofv3 summary -> what bench.py joins into roofline_default_schedule):
    run on a committed trace, every matrix-core instantiation gets a bench key with calls per step and an average duration, and the summed
    kernel time is the CSV's total."""
    import json
    import subprocess
    src = sorted(glob.glob(os.path.join(ROOT, "profiles", "round4_*_default_kernel_stats.csv")))
    assert src, "no committed default-schedule trace"
    out = tmp_path / "ds.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "default_schedule.py"), str(out), src[0], "--workload", "c3|b64|f32", "--steps-in-trace", "7"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    assert d["workload_tag"] == "c3|b64|f32" and d["steps_in_trace"] == 7 and len(d["source_sha16"]) == 16
    total = 0.0
    with open(src[0]) as fh:
        for row in csv.DictReader(fh):
            total += float(row["TotalDurationNs"])
    assert abs(d["summed_kernel_ms_per_step"] - total / 7 / 1e6) < 1e-6
    keys = d["by_bench_key"]
    assert any(k.startswith("k11|d1|") for k in keys) and all(v["calls_per_step"] > 0 and v["avg_us"] > 0 for v in keys.values())
