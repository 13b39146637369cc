"""Per-kernel totals of the fit's step chain from a rocprofv3 kernel-stats CSV: scratch/fit_kernel_times.py <run_kernel_stats.csv>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_fs_" in n:
        print("%-18s calls %4s total %7.2f ms avg %6.1f us max %6.1f us" % (n.split("k_fs_")[1].split("(")[0].split("<")[0], r["Calls"],
              int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
