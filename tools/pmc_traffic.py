"""HBM-side traffic per kernel launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py).

Run ON THE GPU BOX, each counter in its own pass (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dtoid
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dtoid
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per
128-B request of a wide read, i.e. HALF the bytes -> doubled here; WRITE_SIZE is exact for 16-B-per-lane stores.
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (mean over the profiled launches). The counters sit on
the L2's fabric side, so Infinity-Cache hits are included: this is traffic beyond L2, an upper bound on HBM bytes.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        short = name.split("::")[-1].split("(")[0] if "::" in name else name.split("(")[0]
        acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"fetch_size_kib": f, "write_size_kib": w, "traffic_bytes": (2.0 * f + w) * 1024.0}
    json.dump({"note": "per launch; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024, see tools/pmc_traffic.py",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print("%-40s %12.1f MB" % (k, v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main()
