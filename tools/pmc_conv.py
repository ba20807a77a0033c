"""FETCH_SIZE per conv_nhwc launch of tools/conv_ab.py, grouped per shape (13 launches each, in order).
  python tools/pmc_conv.py <dir of rocprofv3 --pmc FETCH_SIZE run>"""
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == sys.argv[2] and "conv_nhwc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
per = 13
for i in range(0, len(rows), per):
    g = rows[i:i + per]
    v = [float(r["Counter_Value"]) for r in g]
    print(i // per, g[0]["Kernel_Name"][40:75], "grid", g[0].get("Grid_Size", "?"), "mean %s %.1f" % (sys.argv[2], sum(v) / len(v)))
