"""Where the waves of each kernel spend their cycles (MI355X_MICROARCH.md, SQ counters):
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d out --output-format csv -- python3 <script>
  python tools/pmc_wave_breakdown.py out
WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing; shares of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1][:40] + (" " + name[name.find("<"):name.find(">") + 1][:34] if "<" in name else "")
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-76s %10s %8s %8s %8s %8s %10s" % ("kernel", "wave Mcyc", "parked", "stalled", "issuing", "lds-iss", "mfma/busy"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    w = v.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0:
        continue
    busy = v.get("SQ_BUSY_CYCLES", 0.0)
    print("%-76s %10.1f %8.3f %8.3f %8.3f %8.3f %10s" % (
        k, w / 1e6, v.get("SQ_WAIT_ANY", 0) / w, v.get("SQ_WAIT_INST_ANY", 0) / w, v.get("SQ_ACTIVE_INST_ANY", 0) / w,
        v.get("SQ_WAIT_INST_LDS", 0) / w, ("%.3f" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy)) if busy else "-"))
