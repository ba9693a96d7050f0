"""Developer helper: effective shader clock per kernel = GRBM_GUI_ACTIVE / duration (rocprofv3 counter_collection + kernel_trace CSVs)."""
import csv, sys, collections, glob
d = sys.argv[1]
dur = {}
for row in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
    dur[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for row in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name, ns = dur[row["Dispatch_Id"]]
    a = agg[name]
    a[0] += float(row["Counter_Value"]); a[1] += ns; a[2] += 1
for name, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-60s n=%3d  %.3f GHz  (avg %.3f ms)" % (name.replace("void vits::", "")[:60], n, cyc / ns, ns / n / 1e6))
