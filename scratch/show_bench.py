"""Prints the interesting parts of a bench.py JSON line: scratch/show_bench.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_regions", "pass_mode", "n_gpus")})
print("roofline", d["roofline"])
print("stages", d["stages_ms"], "e2e", d["end_to_end_run"]["seconds"], d["end_to_end_run"]["wall_s"])
for k in ("rccl", "value_per_rank"):
    if k in d:
        print(k, d[k])
for c in d.get("configs", []):
    print(c["workload"][:44], "%.2fe9" % (c["value"] / 1e9), "step %.3f" % c["ms_per_step"], c["stages_ms"],
          {k: round(v, 3) for k, v in c["roofline"].items() if k in ("frac", "frac_step")}, c["end_to_end_run"])
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"])
