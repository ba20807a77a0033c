"""The idle stretches of the MAIN stream inside one finetune step (rocprofv3 --kernel-trace csv): every gap of at least
--min-us between two consecutive main-stream kernels, with the kernel in front of it, the kernel behind it, and what the
other streams ran meanwhile -- i.e. what the critical path was waiting for.
  python tools/step_gaps.py trace.csv [--min-us 40]"""
import argparse
import collections
import csv
import re


def short(n):
    return re.sub(r"\(anonymous namespace\)::|^void |at::native::", "", n).split("(")[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--min-us", type=float, default=40.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "amsgrad" in r["Kernel_Name"]]
    fr = rows[idx[-2] + 1: idx[-1] + 1]
    t0 = int(fr[0]["Start_Timestamp"])
    streams = sorted({r["Stream_Id"] for r in fr}, key=int)
    main_s = streams[0]
    mainrows = [r for r in fr if r["Stream_Id"] == main_s]
    others = [r for r in fr if r["Stream_Id"] != main_s]
    span = (int(fr[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mainrows) / 1e3
    print("step span %.0f us; main stream %s busy %.0f us in %d launches; other streams %s" % (span, main_s, busy, len(mainrows), streams[1:]))
    total = 0.0
    for p, q in zip(mainrows[:-1], mainrows[1:]):
        g0, g1 = int(p["End_Timestamp"]), int(q["Start_Timestamp"])
        gap = (g1 - g0) / 1e3
        if gap < a.min_us:
            continue
        total += gap
        during = collections.Counter()
        for r in others:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            ov = min(e, g1) - max(s, g0)
            if ov > 0:
                during["s%s %s" % (r["Stream_Id"], short(r["Kernel_Name"]))] += ov / 1e3
        print("at %7.0f us  gap %6.0f us  after %-44s before %-44s | %s" % (
            (g0 - t0) / 1e3, gap, short(p["Kernel_Name"]), short(q["Kernel_Name"]),
            "; ".join("%s %.0f" % kv for kv in during.most_common(3)) or "(nothing on the other streams)"))
    print("gaps >= %.0f us: %.0f us in all" % (a.min_us, total))


if __name__ == "__main__":
    main()
