"""rocprofv3 --kernel-trace --stats leaves a rocpd sqlite database on this image: print its `top_kernels` view as CSV
(Name,Calls,TotalDurationUs,AverageUs,Percentage).  usage: python tools/kstats.py <dir or .db> > profiles/rNN_x.csv"""
import glob, os, sqlite3, sys
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
con = sqlite3.connect(dbs[-1])
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for name, calls, tot, avg, pct in rows:
    print(f'"{name}",{calls},{tot:.3f},{avg:.3f},{pct}')
