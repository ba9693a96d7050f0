"""Shared helpers of the PMC reducers (tools/pmc_traffic.py, tools/pmc_mfma.py): rocprofv3 kernel names -> the kernel keys
bench.py prints (`k<taps>|d<dilation>|t<tile>|e<epilogue>`), and the source hash that ties a profile artefact to a build."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (WM, WN, MR, NR) of conv_mfma_kernel -> ConvTile index (kernels.h)
TILES = {(2, 2, 2, 2): 0, (1, 4, 2, 2): 1, (1, 4, 1, 2): 2, (1, 4, 2, 1): 3, (1, 4, 1, 1): 4, (4, 1, 1, 1): 5}
_CONV = re.compile(r"conv_mfma_kernel<(-?\d+), (-?\d+), (true|false), (\d+), (\d+), (\d+), (\d+), (\d+)>")
_CONV16 = re.compile(r"conv16_kernel<(-?\d+), (-?\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+)>")
_WAVENET16 = re.compile(r"wavenet16_kernel<(\d+), (\d+), (\w+)(?:, \d+)?>")
_COUPLE16 = re.compile(r"flow_couple16_kernel<(\w+)(?:, \d+)*>")
_WAVENET32 = re.compile(r"wavenet32_kernel<(\d+), (\d+)>")
# rbblock16_kernel<KT, C, NSTRIP, NRW, MRW, D0, D1, D2, BF[, STREAM]> (block-shape parameters between C and the dilations: any number of them)
_RBBLOCK16 = re.compile(r"rbblock16_kernel<(-?\d+), (\d+)(?:, \d+)*?, (\d+), (\d+), (\d+), (true|false)(?:, (?:true|false))?>")
_SPLIT = re.compile(r"conv_split_kernel<(-?\d+), (-?\d+), (\d+)>")
_CONVT16 = re.compile(r"convt16_kernel<(\d+), (\d+), (\d+), (\w+)>")
_CONVT16L = re.compile(r"convt16_lines_kernel<(\d+), (\w+)>")
_RBBLOCK32 = re.compile(r"rbblock32_kernel<(\d+), (\d+)>")
_GROUP = re.compile(r"conv_group_kernel<(-?\d+)>")
_RBPAIR32 = re.compile(r"rbpair32_kernel<(-?\d+), (-?\d+), (\d+)>")
_RBPAIR16 = re.compile(r"rbpair16_kernel<(-?\d+), (-?\d+), (\d+), (\d+), (\w+)(?:, \w+)?>")
# conv16.hip: (WM, WN, MR, NR) -> tile index; Epi16 -> the epilogue tag the engine prints
TILES16 = {(2, 2, 2, 4): 0, (1, 4, 2, 2): 1, (1, 4, 1, 2): 2, (1, 4, 2, 1): 3, (1, 4, 1, 1): 4, (2, 2, 2, 2): 5, (4, 1, 1, 4): 6}
EPI16 = {0: "e0", 1: "e1", 2: "e2", 3: "e0g", 4: "e2g"}


def bench_key(kernel_name):
    """`void vits::conv_mfma_kernel<11, 1, true, 2, 2, 2, 2, 0>(vits::ConvParams)` -> `k11|d1|t0|e0` (None for other kernels)"""
    m = _CONV16.search(kernel_name)
    if m:  # `conv16_kernel<11, 1, 4, 1, 1, 4, 3, false>` -> `k11|d1|T6|e0g`
        kt, dil, wm, wn, mr, nr, epi, _ = m.groups()
        tile = TILES16.get((int(wm), int(wn), int(mr), int(nr)))
        return None if tile is None else f"k{kt}|d{dil}|T{tile}|{EPI16.get(int(epi), 'e?')}"
    m = _RBPAIR16.search(kernel_name)
    if m:  # `rbpair16_kernel<11, 1, 64, 2, false>` -> `k11|d1|F64|e0g`
        kt, dil, c = m.groups()[:3]
        return f"k{kt}|d{dil}|F{c}|e0g"
    m = _WAVENET16.search(kernel_name)
    if m:  # `wavenet16_kernel<192, 5, false>` -> `k5|d1|W192|e1`
        h, kt, _ = m.groups()
        return f"k{kt}|d1|W{h}|e1"
    m = _COUPLE16.search(kernel_name)
    if m:  # `flow_couple16_kernel<false>` -> `k5|d1|C192|e1` (one whole coupling layer of the flow)
        return "k5|d1|C192|e1"
    m = _WAVENET32.search(kernel_name)
    if m:  # `wavenet32_kernel<192, 5>` -> `k5|d1|w192|e1`
        h, kt = m.groups()
        return f"k{kt}|d1|w{h}|e1"
    m = _RBPAIR32.search(kernel_name)
    if m:  # `rbpair32_kernel<11, 1, 64>` -> `k11|d1|f64|e0`
        kt, dil, c = m.groups()
        return f"k{kt}|d{dil}|f{c}|e0"
    m = _RBBLOCK16.search(kernel_name)
    if m:  # `rbblock16_kernel<11, 32, 4, 3, 1, 1, 3, 5, false>` -> `k11|d135|B32|e0g`
        kt, c, d0, d1, d2, _ = m.groups()
        return f"k{kt}|d{d0}{d1}{d2}|B{c}|e0g"
    m = _SPLIT.search(kernel_name)
    if m:  # `conv_split_kernel<11, 1, 4>` -> `k11|d1|S128|e0` (VITS_ARITH_F32_SPLIT: 128-row tiles; the engine prints the same tag)
        kt, dil, wm = m.groups()
        return f"k{kt}|d{dil}|S128|e0" if int(wm) == 4 else None
    m = _RBBLOCK32.search(kernel_name)
    if m:  # `rbblock32_kernel<32, 2>` -> `k3|d135|b32|e0` (whole 3-tap resblock, fp32)
        return f"k3|d135|b{m.group(1)}|e0"
    m = _CONVT16.search(kernel_name)
    if m:  # `convt16_kernel<4, 1, 16, false>` -> `k2|d-1|S4.1.16|e2g` (the engine prints the same tag: kernels.h convt16_stream_tag)
        nr, cs, rs, _ = m.groups()
        return f"k2|d-1|S{nr}.{cs}.{rs}|e2g"
    m = _CONVT16L.search(kernel_name)
    if m:  # `convt16_lines_kernel<128, false>` -> `k2|d-1|SL128|e2g`
        return f"k2|d-1|SL{m.group(1)}|e2g"
    m = _GROUP.search(kernel_name)
    if m:  # `conv_group_kernel<3>` -> `kG|d3|G0|e0` (two or three member convolutions of 11 / 7 / 3 taps per launch, 128 x 128 tile)
        return f"kG|d{m.group(1)}|G0|e0"
    m = _CONV.search(kernel_name)
    if not m:
        return None
    kt, dil, _, wm, wn, mr, nr, epi = m.groups()
    tile = TILES.get((int(wm), int(wn), int(mr), int(nr)))
    if tile is None:
        return None
    return f"k{kt}|d{dil}|t{tile}|e{epi}"


def short_name(kernel_name):
    s = kernel_name.replace("void vits::", "").replace("(vits::ConvParams)", "")
    return s if "<" in s else s.split("(")[0]


def source_sha16():
    spec = importlib.util.spec_from_file_location("vits_cpp_amd", os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
    pkg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pkg)
    return pkg.source_sha16()


def rebuild_by_key(kernels):
    """by_bench_key of an artefact from its `kernels` entries (the instantiation with the most sampled launches wins a key): what the reducers
    compute at collection time, and what `--rekey FILE...` recomputes when a kernel's template signature (and with it the regex above) changed
    after the counters were collected."""
    by_key = {}
    for name, e in kernels.items():
        key = bench_key(name)
        if key and (key not in by_key or e.get("launches_sampled", 0) > by_key[key].get("launches_sampled", 0)):
            by_key[key] = dict(e, kernel_name=name)
    return by_key


if __name__ == "__main__":
    import json, sys
    if len(sys.argv) > 2 and sys.argv[1] == "--rekey":
        for path in sys.argv[2:]:
            d = json.load(open(path))
            n0 = len(d.get("by_bench_key", {}))
            d["by_bench_key"] = rebuild_by_key(d["kernels"])
            json.dump(d, open(path, "w"), indent=1, sort_keys=True)
            print("rekeyed", path, n0, "->", len(d["by_bench_key"]), "keys")
