"""Bytes beyond L2 and launches per call of the DTOID bench legs from rocprofv3 PMC passes over tools/dtoid_leg.py:
    python tools/pmc_dtoid_traffic.py <dir with <leg>_f/ and <leg>_w/ sub-directories> <calls incl. the first> [out.json]
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (gfx950: FETCH_SIZE sees 64 of the 128 bytes of a wide read; Infinity-Cache hits are
counted: an upper bound on HBM bytes), summed over every kernel of the run, divided by the number of calls. The first call also
fills the template cache / records the launch plans: a few per cent more launches than a steady-state call."""
import collections
import csv
import glob
import json
import os
import sys


def total(d, counter):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        return None, 0
    tot, n = 0.0, 0
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        tot += float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or r.get("Correlation_Id")
        if key not in seen:
            seen.add(key)
            n += 1
    return tot, n


def main():
    root, calls = sys.argv[1], float(sys.argv[2])
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/dtoid_leg.py (eager launches), "
                     "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per call, all kernels; calls per run = %d" % calls, "legs": {}}
    for leg in ("forward", "forward_batch", "forward_pairs", "finetune"):
        f, nf = total(os.path.join(root, leg + "_f"), "FETCH_SIZE")
        w, nw = total(os.path.join(root, leg + "_w"), "WRITE_SIZE")
        if f is None or w is None:
            continue
        out["legs"][leg] = {"traffic_bytes_per_call": (2.0 * f + w) * 1024 / calls, "read_bytes_per_call": 2.0 * f * 1024 / calls,
                            "write_bytes_per_call": w * 1024 / calls, "launches_per_call": round(max(nf, nw) / calls, 1)}
        print(leg, json.dumps(out["legs"][leg]))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
