"""Developer helper: summarise rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches)."""
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            short = name.replace("void vits::", "").replace("(vits::ConvParams)", "")
            if "conv_mfma_kernel" not in short:
                short = short.split("(")[0]
            agg[short][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (f, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[short] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("SQ_BUSY_CYCLES", 0)))
ctrs = sorted({c for n in agg for c in agg[n]})
print("kernel," + ",".join(ctrs))
for n in names[:28]:
    print(n + "," + ",".join("%.4g" % agg[n].get(c, 0) for c in ctrs))
