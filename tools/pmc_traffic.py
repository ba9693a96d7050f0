"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate passes) into a JSON of HBM bytes per
launch for every kernel name, following /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE on
gfx950 under-reports and must be CALIBRATED on a known byte count in the kernel's own access pattern. The calibration
factors come from tools/fetch_calib.hip (pure streaming kernels over a 1 GiB buffer, well past the 256 MiB Infinity Cache):
pass them as --fetch-cal / --write-cal (defaults: the values measured in profiles/round2_fetch_calibration.json).

usage: pmc_traffic.py OUT.json [--fetch-cal F] [--write-cal W] CSV_GLOB..."""
import collections, csv, glob, json, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_common import bench_key, source_sha16

# --workload TAG: which bench workload the passes ran (bench.py only uses an artefact whose tag equals its own: the bytes and
# busy fractions of a kernel instantiation depend on the shapes it was launched with)
WORKLOAD_TAG = "c3|b64|f32"
if "--workload" in sys.argv:
    i = sys.argv.index("--workload")
    WORKLOAD_TAG = sys.argv[i + 1]
    del sys.argv[i:i + 2]

# --steps-in-trace N: bench steps the traced command ran (warm-up + timed: with --single-pass every one of them is instrumented), for
# the whole-step total; --samples-per-step S is recorded beside it (bench.py checks it against its own run)
STEPS_IN_TRACE, SAMPLES_PER_STEP = 0, None
for flag in ("--steps-in-trace", "--samples-per-step"):
    if flag in sys.argv:
        i = sys.argv.index(flag)
        if flag == "--steps-in-trace":
            STEPS_IN_TRACE = int(sys.argv[i + 1])
        else:
            SAMPLES_PER_STEP = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]

argv = sys.argv[1:]
out = argv.pop(0)
FETCH_CAL, WRITE_CAL = 2.0, 1.0
while argv and argv[0].startswith("--"):
    flag = argv.pop(0)
    val = float(argv.pop(0))
    if flag == "--fetch-cal":
        FETCH_CAL = val
    elif flag == "--write-cal":
        WRITE_CAL = val
fetch, write, calls_f, calls_w = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for pattern in argv:
    for f in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if row["Counter_Name"] == "FETCH_SIZE":
                fetch[name] += float(row["Counter_Value"]); calls_f[name] += 1
            elif row["Counter_Name"] == "WRITE_SIZE":
                write[name] += float(row["Counter_Value"]); calls_w[name] += 1
res, by_key = {}, {}
for name in set(fetch) | set(write):
    raw = 1024.0 * fetch[name] / max(calls_f[name], 1)
    fb = FETCH_CAL * raw
    wraw = 1024.0 * write[name] / max(calls_w[name], 1)
    wb = WRITE_CAL * wraw
    res[name] = {"fetch_raw_bytes_per_launch": raw, "fetch_bytes_per_launch": fb, "write_raw_bytes_per_launch": wraw, "write_bytes_per_launch": wb,
                 "hbm_bytes_per_launch": fb + wb, "launches_sampled": max(calls_f[name], calls_w[name])}
    key = bench_key(name)
    if key and (key not in by_key or res[name]["launches_sampled"] > by_key[key]["launches_sampled"]):
        by_key[key] = dict(res[name], kernel_name=name)
whole = None
if STEPS_IN_TRACE > 0:
    # every kernel of the trace (library kernels AND the few torch / RCCL ones): counter totals / steps
    fb = FETCH_CAL * 1024.0 * sum(fetch.values()) / STEPS_IN_TRACE
    wb = WRITE_CAL * 1024.0 * sum(write.values()) / STEPS_IN_TRACE
    whole = {"fetch_bytes_per_step": fb, "write_bytes_per_step": wb, "hbm_bytes_per_step": fb + wb, "steps_in_trace": STEPS_IN_TRACE,
             "samples_per_step": SAMPLES_PER_STEP, "launches_per_step": sum(calls_f.values()) / STEPS_IN_TRACE}
cal = f"rocprofv3 --pmc FETCH_SIZE x {FETCH_CAL:g} + WRITE_SIZE x {WRITE_CAL:g} (KiB -> bytes; factors calibrated with tools/fetch_calib.hip)"
json.dump({"note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-pass`",
           "calibration": cal, "source_sha16": source_sha16(), "workload_tag": WORKLOAD_TAG, "whole_step": whole, "by_bench_key": by_key, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
print(len(res), "kernels ->", out)
