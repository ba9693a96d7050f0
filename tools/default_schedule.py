"""rocprofv3 --kernel-trace --stats summary of the LIBRARY-DEFAULT schedule (bench.py --no-prof: no per-kernel events, the three resblocks
of a vocoder stage on three streams, separate launches) -> profiles-ready JSON that bench.py joins with its own algorithmic FLOP
accounting into the `roofline_default_schedule` block: per kernel instantiation the average launch duration and the
calls per step, and the summed kernel time per step (against the wall time it gives the overlap the streams buy).

usage: default_schedule.py OUT.json KERNEL_STATS.csv --workload TAG --steps-in-trace N"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_common import bench_key, source_sha16

args = sys.argv[1:]
tag, steps = "c3|b64|f32", 1
for flag in ("--workload", "--steps-in-trace"):
    if flag in args:
        i = args.index(flag)
        if flag == "--workload":
            tag = args[i + 1]
        else:
            steps = int(args[i + 1])
        del args[i:i + 2]
out, path = args
kern, by_key, total_ns = {}, {}, 0.0
for row in csv.DictReader(open(path)):
    name, calls, tot = row["Name"], int(row["Calls"]), float(row["TotalDurationNs"])
    total_ns += tot
    e = {"calls_per_step": calls / steps, "avg_us": tot / calls / 1e3, "ms_per_step": tot / steps / 1e6}
    kern[name] = e
    key = bench_key(name)
    if key:
        k = by_key.setdefault(key, {"calls_per_step": 0.0, "ms_per_step": 0.0, "kernel_names": []})
        k["calls_per_step"] += e["calls_per_step"]
        k["ms_per_step"] += e["ms_per_step"]
        k["kernel_names"].append(name)
for k in by_key.values():
    k["avg_us"] = 1e3 * k["ms_per_step"] / k["calls_per_step"]
json.dump({"note": "rocprofv3 --kernel-trace --stats of `python3 bench.py --no-prof --no-cpu-baseline --no-sub-results --no-extra-passes` (library default schedule)",
           "source_sha16": source_sha16(), "workload_tag": tag, "steps_in_trace": steps, "summed_kernel_ms_per_step": total_ns / steps / 1e6,
           "by_bench_key": by_key, "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
print(len(kern), "kernels,", round(total_ns / steps / 1e6, 3), "ms of kernels per step ->", out)
