"""Traffic beyond L2 of the finetune step by kernel family, from two rocprofv3 PMC passes (units / corrections: tools/pmc_traffic.py):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 tools/bench_finetune.py --reps 2 --no-graph
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 tools/bench_finetune.py --reps 2 --no-graph
  python tools/pmc_step_traffic.py out/f out/w [steps_profiled]
Prints, per kernel name, launches and GB moved per step (all profiled launches / steps; the warm-up steps count too)."""
import collections
import csv
import glob
import sys


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        short = (name.split("::")[-1].split("(")[0] if "::" in name else name.split("(")[0])[:44]
        tot[short] += float(r["Counter_Value"])
        cnt[short] += 1
    return tot, cnt


fetch, n = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = sorted(((2.0 * fetch.get(k, 0) + write.get(k, 0)) * 1024 / steps, k) for k in set(fetch) | set(write))
print("%-46s %9s %10s %10s %10s" % ("kernel", "launches", "read GB", "write GB", "total GB"))
for t, k in reversed(rows[-22:]):
    print("%-46s %9.0f %10.3f %10.3f %10.3f" % (k, n.get(k, 0) / steps, 2.0 * fetch.get(k, 0) * 1024 / steps / 1e9,
                                                write.get(k, 0) * 1024 / steps / 1e9, t / 1e9))
print("%-46s %9s %10s %10s %10.3f" % ("all kernels", "", "", "", sum(t for t, _ in rows) / 1e9))
