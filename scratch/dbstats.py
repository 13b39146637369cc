"""Per-kernel totals from a rocprofv3 rocpd sqlite database: python scratch/dbstats.py <dir-or-db>"""
import glob, os, sqlite3, sys
p = sys.argv[1]
db = p if p.endswith(".db") else sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, sym))
for r in c.execute(q):
    print("%-56s n=%6d tot=%9.3f ms avg=%9.1f us max=%9.1f us" % (r[0][:56], r[1], r[2], r[3], r[4]))
