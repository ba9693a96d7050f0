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
