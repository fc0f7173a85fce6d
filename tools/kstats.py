"""Dev tool: print a rocprofv3 kernel_stats.csv (newest under the given directory) as per-forward rows.
    python tools/kstats.py gpurun_out/prof_b1 [forwards]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
tot = 0.0
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("void ", "").replace("pio::", "").split("(")[0][:60]
    ms = float(r["TotalDurationNs"]) / 1e6
    tot += ms
    print(f"{name:60s} {int(r['Calls'])/n:8.1f}/fwd {float(r['AverageNs'])/1e3:8.1f} us {ms/n*1e3:9.1f} us/fwd")
print(f"total {tot/n:.3f} ms/fwd")
