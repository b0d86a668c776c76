"""dev: print a rocprofv3 kernel_stats.csv compactly.  usage: kstats.py <dir-or-csv>"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else (glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True) or [None])[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"<.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    print(f"{n:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
