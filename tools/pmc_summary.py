"""Mean of each counter per kernel over the dispatches of a rocprofv3 --pmc run:  python tools/pmc_summary.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:90], r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
