"""Copies the judged summaries of a profile set (scratch/profile_round.sh <tag> on the GPU box, merged back under
gpurun_out/<tag>/) into profiles/ as <name>_*: usage  python3 scratch/archive_profiles.py <tag> <name>"""
import csv, collections, glob, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
FILL, PRED = "k_fill3", "k_predict_rows"


def first(pattern):
    return glob.glob(os.path.join(src, pattern))[0]


bench = json.load(open(os.path.join(src, "bench.json")))
shutil.copy(os.path.join(src, "bench.json"), "profiles/%s_bench.json" % name)
for cfg in ("C3", "C4", "C5", "C2h", "C2t"):
    if os.path.exists(os.path.join(src, "bench_%s.json" % cfg)):
        shutil.copy(os.path.join(src, "bench_%s.json" % cfg), "profiles/%s_bench_%s.json" % (name, cfg))
shutil.copy(first("prof/*kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % name)
rows = [l for l in open(first("prof/*kernel_trace.csv")) if l.startswith('"Kind"') or FILL in l or PRED in l]
open("profiles/%s_kernel_trace_fill_predict.csv" % name, "w").writelines(rows)
for sub in ("e2e_c2", "e2e_c5", "dyn"):
    shutil.copy(first(sub + "/*kernel_stats.csv"), "profiles/%s_%s_kernel_stats.csv" % (name, sub))
    lines = [l for l in open(os.path.join(src, sub + ".log")) if not l.startswith(("E2026", "W2026", "I2026")) and "rocprof" not in l]
    open("profiles/%s_%s.txt" % (name, sub), "w").writelines(lines)
big = 0
out = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", None), ("f64", "F64")):
    f = first(sub + "/*counter_collection.csv")
    rs = [r for r in csv.DictReader(open(f)) if FILL in r["Kernel_Name"] or PRED in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rs if FILL in r["Kernel_Name"])
    rs = [r for r in rs if int(r["Grid_Size"]) >= big // 2]           # the timed launches (not the small end-to-end ones)
    with open("profiles/%s_pmc_%s.csv" % (name, (ctr or "sq").lower()), "w") as g:
        w = csv.writer(g)
        w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        for r in rs:
            w.writerow([r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rs:
        agg[r["Kernel_Name"][:12]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
fill = [v for k, v in out.items() if FILL in k][0]
traffic = fill["FETCH_SIZE"] * 1024 * 2 + fill["WRITE_SIZE"] * 1024
json.dump({"kernel": FILL, "bytes_per_launch": traffic, "fetch_size_kib": fill["FETCH_SIZE"], "write_size_kib": fill["WRITE_SIZE"],
           "lib_sha16": open(os.path.join(src, "lib_sha16")).read().strip(), "build": name,
           "note": "C2, F=100000 (6.4e6 landmark vectors per launch); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                   "128-B requests at 64 B); WRITE_SIZE exact = the sparse rows (nnz + 1.95 entries x 12 B per vector); "
                   "bench.py reports this figure only while the library's hash matches"},
          open("profiles/pmc_traffic.json", "w"), indent=1)
ions = bench["config"]["frames_per_gpu"] * bench["config"]["n_mobile"]
# the FP64-VALU side of the roofline: instructions per landmark vector, the FP64 arithmetic among
# them, and the share of the kernel's cycles in which a SIMD issues a vector instruction (SQ_ACTIVE_INST_VALU counts
# quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE counts the kernel's cycles on each of the 8 XCDs)
if "SQ_INSTS_VALU" in fill and "GRBM_GUI_ACTIVE" in fill:
    floor = sum(fill.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    json.dump({"kernel": FILL, "config": "C2", "insts_per_ion": round(fill["SQ_INSTS_VALU"] / ions, 2),
               "floor_insts_per_ion": round(floor / ions, 2), "salu_per_ion": round(fill["SQ_INSTS_SALU"] / ions, 2),
               "issue_frac": round(fill["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (fill["GRBM_GUI_ACTIVE"] / 8.0), 3),
               "wait_frac": round(fill["SQ_WAIT_ANY"] / fill["SQ_WAVE_CYCLES"], 3),
               "lib_sha16": open(os.path.join(src, "lib_sha16")).read().strip(), "build": name,
               "note": "wave instructions per (frame, ion); floor = FP64 add / mul / fma / transcendental instructions; "
                       "issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs"},
              open("profiles/pmc_valu.json", "w"), indent=1)
for extra in ("ab_fill_c2.txt", "ab_fill_c5.txt", "stages.txt", "bench_tcp2.json", "bench_tcp2_C4.json", "bench_tcp2_C5.json", "e2e_walls_c2.txt",
              "phase_times.txt", "mem_counters.txt", "predict_counters.txt", "ab_predict_c2.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), "profiles/%s_%s" % (name, extra))
print("bench value %.4g lvec/s, fill %.4f ms, predict %.4f ms, frac %.4f, frac_step %.4f" % (
    bench["value"], bench["stages_ms"]["fill"], bench["stages_ms"]["predict"], bench["roofline"]["frac"], bench["roofline"]["frac_step"]))
print("cpu", bench.get("cpu_baseline"))
print("e2e", bench["end_to_end_run"])
print("traffic GB %.4f (fetch x2 %.4f + write %.4f) vs algorithmic %.4f" % (traffic / 1e9, fill["FETCH_SIZE"] * 2048 / 1e9, fill["WRITE_SIZE"] * 1024 / 1e9, ions * 232 / 1e9))
for k, v in out.items():
    print(k, {c: round(x / ions, 2) for c, x in v.items() if c.startswith("SQ")})
