"""print the per-kernel table of the last bench.py run (bench_detail.json)"""
import json, os, sys
d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", sys.argv[1] if len(sys.argv) > 1 else "bench_detail.json")))
print(d.get("config", {}).get("workload"), "ms/step", round(d["ms_per_step"], 3), "instrumented", round(d.get("instrumented_ms_per_step", 0), 3), "kernel ms", round(d.get("kernel_time_ms_per_step", 0), 3))
for k in d.get("top_kernels", [])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-30s %7.3f ms/step %5.1f calls  %6.1f TF" % (k["kernel"], k["ms_per_step"], k["calls_per_step"], k["tflops"] or 0))
