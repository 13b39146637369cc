"""Per-kernel summary of the fit kernels in a rocprofv3 kernel trace: scratch/fit_trace_stats.py <run_kernel_trace.csv>"""
import csv, collections, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
byk = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'k_fs_' in n:
        byk[n.split('k_fs_')[1].split('(')[0].split('<')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
allv = sorted(x for v in byk.values() for x in v)
print("fit span %.1f ms, %d launches" % ((allv[-1][0] + allv[-1][1] - allv[0][0]) / 1e6, len(allv)))
for k, v in byk.items():
    d = np.array([x[1] for x in v])
    print("%-10s n %4d sum %6.1f ms  median %6.1f us  p90 %6.1f  max %6.1f" % (k, len(d), d.sum() / 1e6, np.median(d) / 1e3, np.percentile(d, 90) / 1e3, d.max() / 1e3))
