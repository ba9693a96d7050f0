"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate passes) into a JSON of HBM bytes per
launch for every kernel name, following /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE on
gfx950 under-reports and must be CALIBRATED on a known byte count in the kernel's own access pattern (the guide's exact x2
holds for 16 B/lane streams; these kernels stage with 4 B/lane loads). Calibration used here: launches whose read volume is
known exactly (conv_mfma_kernel<3, 3, false, 1, 4, 1, 1, 0>: one 474.1 MB input tensor, no residual, negligible weights)
report 415 MB raw -> factor 1/0.875 = 1.143 on FETCH_SIZE; WRITE_SIZE matches the known output bytes to <1 % (factor 1)."""
import collections, csv, glob, json, sys
out = sys.argv[1]
FETCH_CAL = 1.0 / 0.875
fetch, write, calls_f, calls_w = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for pattern in sys.argv[2:]:
    for f in glob.glob(pattern):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if row["Counter_Name"] == "FETCH_SIZE":
                fetch[name] += float(row["Counter_Value"]); calls_f[name] += 1
            elif row["Counter_Name"] == "WRITE_SIZE":
                write[name] += float(row["Counter_Value"]); calls_w[name] += 1
res = {}
for name in set(fetch) | set(write):
    raw = 1024.0 * fetch[name] / max(calls_f[name], 1)
    fb = FETCH_CAL * raw
    wb = 1024.0 * write[name] / max(calls_w[name], 1)
    res[name] = {"fetch_raw_bytes_per_launch": raw, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                 "launches_sampled": max(calls_f[name], calls_w[name])}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over `python3 bench.py --steps 2 --warmup 1`; KiB -> bytes, FETCH_SIZE x 1.143 (calibrated on a launch with known read volume), WRITE_SIZE x 1",
           "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
print(len(res), "kernels ->", out)
