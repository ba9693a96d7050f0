"""Reduces the FETCH_SIZE / WRITE_SIZE passes over tools/bin/fetch_calib into calibration factors (known bytes / reported).
usage: fetch_calib_reduce.py OUT.json CSV_GLOB_OR_DIR..."""
import collections, csv, glob, json, os, sys
KNOWN = {"calib_read_dword": (2 ** 30, 0), "calib_read_dwordx4": (2 ** 30, 0), "calib_read_lds_dma": (2 ** 30, 0),
         "calib_copy_dwordx4": (2 ** 30, 2 ** 30), "calib_write_dword": (0, 2 ** 30)}
files = []
for p in sys.argv[2:]:
    files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(p) else glob.glob(p, recursive=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
res = {}
for k, (rd, wr) in KNOWN.items():
    e = {"known_read_bytes": rd, "known_write_bytes": wr}
    if cnt[k]["FETCH_SIZE"]:
        e["fetch_size_bytes"] = 1024.0 * tot[k]["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"]
        if rd:
            e["fetch_factor"] = rd / e["fetch_size_bytes"]
    if cnt[k]["WRITE_SIZE"]:
        e["write_size_bytes"] = 1024.0 * tot[k]["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"]
        if wr:
            e["write_factor"] = wr / e["write_size_bytes"]
    res[k] = e
    print(k, {a: (round(b, 4) if isinstance(b, float) and b < 100 else b) for a, b in e.items()})
json.dump({"note": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), 1 GiB streams; factor = known bytes / (counter KiB x 1024)",
           "kernels": res}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
