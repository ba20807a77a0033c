"""Bytes beyond L2 and launches per call of the DTOID bench legs from rocprofv3 PMC passes over tools/dtoid_leg.py:
    python tools/pmc_dtoid_traffic.py <dir with <leg>_f/ and <leg>_w/ sub-directories> [out.json]
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (gfx950: FETCH_SIZE sees 64 of the 128 bytes of a wide read; Infinity-Cache hits are
counted: an upper bound on HBM bytes), summed over every kernel of the run's LAST call (between the last two marker launches of
tools/dtoid_leg.py: the first call also fills the template cache, packs weights and records launch plans)."""
import collections
import csv
import glob
import json
import os
import sys


def total(d, counter):
    """(counter total, launches) of the LAST call of the run: the dispatches between the last two marker (spin kernel) launches."""
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        return None, 0
    per = collections.OrderedDict()                 # dispatch id -> [kernel name, counter sum]
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        key = int(r.get("Dispatch_Id") or r.get("Correlation_Id"))
        ent = per.setdefault(key, [r["Kernel_Name"], 0.0])
        ent[1] += float(r["Counter_Value"])
    ids = sorted(per)
    marks = [i for i in ids if "spin" in per[i][0].lower()]
    if len(marks) < 2:
        raise SystemExit("no marker launches in " + fs[0])
    body = [i for i in ids if marks[-2] < i < marks[-1]]
    return sum(per[i][1] for i in body), len(body)


def main():
    root = sys.argv[1]
    calls = 1.0
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/dtoid_leg.py (eager launches), "
                     "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B over every kernel of the LAST (steady-state) call of each run", "legs": {}}
    for leg in ("forward", "forward_batch", "forward_pairs", "finetune"):
        f, nf = total(os.path.join(root, leg + "_f"), "FETCH_SIZE")
        w, nw = total(os.path.join(root, leg + "_w"), "WRITE_SIZE")
        if f is None or w is None:
            continue
        out["legs"][leg] = {"traffic_bytes_per_call": (2.0 * f + w) * 1024 / calls, "read_bytes_per_call": 2.0 * f * 1024 / calls,
                            "write_bytes_per_call": w * 1024 / calls, "launches_per_call": round(max(nf, nw) / calls, 1)}
        print(leg, json.dumps(out["legs"][leg]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
