"""Host-side logic of bench.py and the PMC reducers (no GPU): which profile artefact a run may quote, and the kernel-name -> bench-key map."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profile_artifact_needs_matching_sources_workload_and_kernel(tmp_path, monkeypatch):
    """roofline.traffic / mfma_busy_frac come from separate rocprofv3 --pmc passes: bench.py may only quote an artefact that was collected
    (a) for the library sources of this run, (b) on this workload (shape-dependent numbers) and (c) that has the kernel at all —
    otherwise the fields stay null (a hard-coded file name silently went stale)."""
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))

    class Pkg:
        @staticmethod
        def source_sha16():
            return "abc123"

    def write(name, sha, tag, keys):
        (prof / name).write_text(json.dumps({"source_sha16": sha, "workload_tag": tag, "by_bench_key": {k: {"hbm_bytes_per_launch": 1.0} for k in keys}}))

    write("a_pmc_traffic.json", "abc123", "c3|b64|f32", ["k11|d1|t0|e0"])
    write("b_pmc_traffic.json", "zzz", "c3|b64|f32", ["k11|d1|t0|e0"])      # other build
    write("c_pmc_traffic.json", "abc123", "c5|b8|f32", ["k11|d1|t0|e0"])     # other workload
    write("d_pmc_traffic.json", "abc123", "c3|b64|f32", ["k3|d1|t0|e0"])     # kernel missing
    got = bench.find_profile_artifact(Pkg, "_pmc_traffic.json", "k11|d1|t0|e0", "c3|b64|f32")
    assert got is not None and os.path.basename(got[0]) == "a_pmc_traffic.json"
    assert bench.find_profile_artifact(Pkg, "_pmc_traffic.json", "k11|d1|t0|e0", "c3|b1|f32") is None
    assert bench.find_profile_artifact(Pkg, "_pmc_traffic.json", "k7|d1|t0|e0", "c3|b64|f32") is None
    assert bench.find_profile_artifact(Pkg, "_pmc_mfma.json", "k11|d1|t0|e0", "c3|b64|f32") is None


def test_kernel_names_map_to_the_keys_the_engine_prints():
    """tools/pmc_common.bench_key turns a rocprofv3 kernel name into the `k|d|tile|epilogue` key of the engine's profiler entries; every
    kernel family of the library has to map (an unmapped family silently loses its PMC numbers)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_common
    finally:
        sys.path.pop(0)
    cases = {
        "void vits::conv_mfma_kernel<11, 1, true, 2, 2, 2, 2, 0>(vits::ConvParams)": "k11|d1|t0|e0",
        "void vits::conv_mfma_kernel<1, 1, true, 4, 1, 1, 1, 0>(vits::ConvParams)": "k1|d1|t5|e0",
        "void vits::conv16_kernel<11, 1, 4, 1, 1, 4, 3, false>(vits::Conv16Params)": "k11|d1|T6|e0g",
        "void vits::rbpair16_kernel<11, 5, 128, 4, false, true>(vits::RbPairParams)": "k11|d5|F128|e0g",
        "void vits::rbpair16_kernel<7, 1, 64, 2, true, false>(vits::RbPairParams)": "k7|d1|F64|e0g",
        "void vits::rbpair32_kernel<3, 5, 32>(vits::RbPair32Params)": "k3|d5|f32|e0",
        "void vits::wavenet32_kernel<192, 5>(vits::WaveNet32Params)": "k5|d1|w192|e1",
        "void vits::wavenet16_kernel<192, 5, false>(vits::WaveNet16Params)": "k5|d1|W192|e1",
        "void vits::wavenet16_kernel<192, 5, true, 1>(vits::WaveNet16Params)": "k5|d1|W192|e1",
        "void vits::flow_couple16_kernel<false>(vits::FlowCouple16Params)": "k5|d1|C192|e1",
        "void vits::conv_group_kernel<3>(vits::ConvGroupParams)": "kG|d3|G0|e0",
        "void vits::rbblock16_kernel<11, 64, 1, 3, 5, false>(vits::RbBlockParams)": "k11|d135|B64|e0g",
    }
    for name, key in cases.items():
        assert pmc_common.bench_key(name) == key, name
    assert pmc_common.bench_key("vits::add_layer_norm_kernel(float const*, long)") is None


def test_final_line_is_small_and_parses(tmp_path, monkeypatch, capsys):
    """The driver parses the LAST stdout line of bench.py and keeps a bounded tail (round 4: a 24 KB line came back as `parsed: null`).
    compact_line() of a worst-case result — the round-4 record with every string padded and 64 sub-results' worth of prose — stays under
    4 KB, is one line of valid JSON, and carries the contract's keys with `roofline` and `cpu_baseline`; the full record goes to a side file."""
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test2")
    with open(os.path.join(ROOT, "tests", "golden", "bench_record_r4.json")) as fh:
        res = json.load(fh)
    assert len(json.dumps(res)) > 20000  # the round-4 shape
    pad = "x" * 4000
    res["config"]["workload"] += pad
    res["metric"] += pad
    res["roofline"]["kernel"] += pad
    res["roofline"]["traffic_unit"] = pad
    res["cpu_baseline"]["sample"] += pad
    res["top_kernels"] = res["top_kernels"] * 8
    for name, d in res["sub_results"].items():
        if isinstance(d, dict):
            d["schedule"] = pad
            d["hbm_source"] = "profiles/round5_some_quite_long_artefact_name_pmc_traffic.json"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    rel = bench.write_detail(res)
    assert rel == "bench_detail.json" and json.load(open(tmp_path / rel))["top_kernels"]
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))["metric"] == res["metric"]
    line = bench.compact_line(res, rel)
    print("noise before")
    print(line)
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(last.encode()) < 4096 and "\n" not in line
    got = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert got["value"] == float("%.6g" % res["value"]) and got["config"]["source_sha16"] == res["config"]["source_sha16"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(got["roofline"])
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-5
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(got["cpu_baseline"])
    assert set(got["sub_results"]) == {"c2_f32", "c2_f16", "c3_f32", "c3_f16", "c3_bf16", "c5_f32", "c5_bf16"}
    assert all(set(e) >= {"value", "ms_per_step", "frac_of_binding_roof", "hbm_source"} for e in got["sub_results"].values())
    # no prose survives in the sub-results
    assert all(not isinstance(v, str) or len(v) < 100 for e in got["sub_results"].values() for v in e.values())


def test_the_record_states_the_host_cores_and_a_workload_label_that_survives_the_driver():
    """VERDICT r5 weak 7 / 8. (a) `cpu_baseline.cores` is the PHYSICAL core count of the host, with the CPU model beside it, and the oracle's thread count is
    `threads` (round 5 printed the fastest thread count of its sweep as "cores" on a 256-thread host); (b) `config.workload` keeps mode and arithmetic within the
    120 characters the driver's record keeps: at most 110, the two facts first — for every workload the bench can run."""
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test3")
    cpu = bench.host_cpu_info()
    assert cpu["physical_cores"] >= 1 and cpu["hardware_threads"] >= cpu["physical_cores"] and isinstance(cpu["cpu_model"], str) and cpu["cpu_model"]
    assert cpu["hardware_threads"] == (os.cpu_count() or 1) or cpu["hardware_threads"] > 0
    for c5 in (False, True):
        for B, T in ((64, 128), (1, 128), (8, 1024), (512, 2048)):
            for mode in ("reference", "hf"):
                for arith in ("f32", "f16", "bf16"):
                    for pinned in (0, 2):
                        w = bench.workload_label(c5, B, T, mode, arith, pinned)
                        assert len(w) <= 110, w
                        head = w[:40]
                        assert arith in head.split() and (mode[:3] + "-mode") in head.split(), w
    assert bench.workload_label(False, 64, 128, "reference", "f32").startswith("c3 b64x128 ref-mode f32 predicted-dur | ")
    assert bench.workload_label(False, 1, 128, "reference", "f16").startswith("c2 b1x128 ref-mode f16 ")
    # the compact line carries the split cpu_baseline fields and never more than the label's budget
    res = {"metric": "m", "value": 1.0, "config": {"workload": "w" * 500}, "cpu_baseline": {"value": 2.0, "unit": "samples/s", "cores": 128, "threads": 32, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                                                                                             "kind": "port", "host_threads": 256, "sample": "s"}}
    got = json.loads(bench.compact_line(res))
    assert len(got["config"]["workload"]) <= 110
    assert got["cpu_baseline"]["cores"] == 128 and got["cpu_baseline"]["threads"] == 32 and got["cpu_baseline"]["host_threads"] == 256 and "EPYC" in got["cpu_baseline"]["cpu_model"]
