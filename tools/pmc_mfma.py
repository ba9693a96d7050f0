"""MFMA-busy and LDS-bank-conflict summary per kernel from rocprofv3 --pmc passes (counter_collection CSVs) joined with the
kernel trace of the same run (kernel durations).

  pass A: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ...
  pass B: rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS ...

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  — the counter ticks once per cycle a SIMD's matrix
pipe is busy (MI355X_MICROARCH.md: 64 cycles per v_mfma_f32_32x32x2_f32, 32 per 32x32x16 bf16), SQ_BUSY_CU_CYCLES once per
cycle a CU has a wave: the fraction of the time CUs are occupied that their four matrix pipes are issuing. A second figure
normalises by the whole launch instead: busy / (1024 SIMDs x launch duration x 2.4 GHz) (lower bound: the clock under load is
2.15-2.35 GHz in fp32 mode). clock_ghz_estimate = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch duration: the shader clock the
launch actually ran at (fp32 kernels: 2.36-2.42 GHz; MFMA-dense 16-bit kernels: 1.25-1.8 GHz — the power budget; launches under ~100 us
are dominated by the counter window and are not meaningful). mfma_busy_frac_of_elapsed_clocks = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8):
the same fraction against the clocks that really elapsed. lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles lost to conflicts over LDS-busy cycles).

usage: pmc_mfma.py OUT.json DIR_OR_CSV_GLOB...
       pmc_mfma.py --refresh FILE.json...      (recompute the derived fields of an existing artefact from its own counters_per_launch)"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_common import bench_key, source_sha16

# --workload TAG: which bench workload the passes ran (bench.py only uses an artefact whose tag equals its own: the bytes and
# busy fractions of a kernel instantiation depend on the shapes it was launched with)
WORKLOAD_TAG = "c3|b64|f32"
if "--workload" in sys.argv:
    i = sys.argv.index("--workload")
    WORKLOAD_TAG = sys.argv[i + 1]
    del sys.argv[i:i + 2]



def derive(e):
    """Derived fractions of one kernel entry from its per-launch counters (and its average duration, if the trace had it)."""
    per, us = e["counters_per_launch"], e.get("avg_duration_us")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        if per.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy_frac"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * per["SQ_BUSY_CU_CYCLES"])
        if us:
            e["mfma_busy_frac_of_launch_at_2p4ghz"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 1e3 * 2.4)
        if per.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac_of_elapsed_clocks"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0)
    if per.get("GRBM_GUI_ACTIVE") and us:
        e["clock_ghz_estimate"] = per["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)
    if per.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
    return e


if len(sys.argv) > 2 and sys.argv[1] == "--refresh":
    for path in sys.argv[2:]:
        d = json.load(open(path))
        for e in d["kernels"].values():
            derive(e)
        for e in d["by_bench_key"].values():
            derive(e)
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
        print("refreshed", path)
    sys.exit(0)

out = sys.argv[1]
ctr = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))   # kernel -> counter -> dispatch ids
dur_ns = collections.defaultdict(float)
dur_n = collections.Counter()
files = []
for pattern in sys.argv[2:]:
    if os.path.isdir(pattern):
        files += glob.glob(os.path.join(pattern, "**", "*.csv"), recursive=True)
    else:
        files += glob.glob(pattern, recursive=True)
for f in files:
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames or []
    if "Counter_Name" in cols:
        for row in rd:
            ctr[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[row["Kernel_Name"]][row["Counter_Name"]].add((f, row["Dispatch_Id"]))
    elif "Start_Timestamp" in cols and "Kernel_Name" in cols:
        for row in rd:
            dur_ns[row["Kernel_Name"]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            dur_n[row["Kernel_Name"]] += 1
res, by_key = {}, {}
for name, c in ctr.items():
    n = {k: max(len(v), 1) for k, v in ndisp[name].items()}
    per = {k: v / n[k] for k, v in c.items()}
    e = {"counters_per_launch": per, "launches_sampled": max(n.values())}
    if dur_n[name]:
        e["avg_duration_us"] = dur_ns[name] / dur_n[name] / 1e3
    res[name] = derive(e)
    key = bench_key(name)
    if key and (key not in by_key or e["launches_sampled"] > by_key[key]["launches_sampled"]):
        by_key[key] = dict(e, kernel_name=name)
json.dump({"note": "rocprofv3 --kernel-trace --pmc passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-pass`; formulas in tools/pmc_mfma.py",
           "source_sha16": source_sha16(), "workload_tag": WORKLOAD_TAG, "by_bench_key": by_key, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
top = sorted(by_key.items(), key=lambda kv: -kv[1]["counters_per_launch"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["launches_sampled"])[:12]
for k, e in top:
    print(f"{k:18s} mfma_busy {e.get('mfma_busy_frac', float('nan')):.3f}  of-launch@2.4GHz {e.get('mfma_busy_frac_of_launch_at_2p4ghz', float('nan')):.3f}  "
          f"lds_conflict {e.get('lds_bank_conflict_frac', float('nan')):.4f}")
print(len(res), "kernels ->", out)
