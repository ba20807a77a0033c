"""SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel name from a rocprofv3 --pmc run:
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d out --output-format csv -- python3 tools/train_layers_bench.py --what wgrad --only "cf.768|b1.l06|s3" --reps 2
  python tools/pmc_lds_conflicts.py out"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1][:60] + (" " + name[name.find("<"):name.find(">") + 1][:40] if "<" in name else "")
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-90s %14s %14s %8s" % ("kernel", "conflict cycles", "LDS cycles", "share"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if a > 0:
        print("%-90s %14.0f %14.0f %8.3f" % (k, c, a, c / a))
