"""Individual launches of the kernels matching a pattern inside ONE steady-state step of a rocprofv3 --kernel-trace csv
(the step = the launches between the last two occurrences of the marker kernel), with grid size and template arguments:
which layers of the finetune step are the slow ones.
  python tools/trace_detail.py trace.csv --match "wgrad_kernel|conv_nhwc" [--top 50]"""
import argparse
import collections
import csv
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="amsgrad")
    ap.add_argument("--match", default="wgrad_kernel|conv_nhwc_kernel")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) *
                         max(1, int(r.get("Grid_Size_Z", 1) or 1)), int(r.get("Workgroup_Size_X", 256) or 256),
                         int(r.get("LDS_Block_Size", 0) or 0)))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2].lower()]
    pairs = [(a0, a1) for a0, a1 in zip(marks[:-1], marks[1:]) if a1 - a0 > 100]
    step = rows[pairs[-1][0] + 1: pairs[-1][1] + 1]
    sel = [(e - s, i, n, g, w, l) for i, (s, e, n, g, w, l) in enumerate(step) if re.search(a.match, n)]
    groups = collections.defaultdict(list)
    for d, i, n, g, w, l in sel:
        targs = re.search(r"<([^>]*)>", n)
        base = re.sub(r"<.*", "", n).split("::")[-1]
        groups[(base, targs.group(1) if targs else "", g // max(w, 1), l)].append(d)
    out = sorted(((sum(v), len(v), k) for k, v in groups.items()), reverse=True)
    print("total %.3f ms in %d launches" % (sum(d for d, *_ in sel) * 1e-6, len(sel)))
    for tot, cnt, (base, targs, wgs, lds) in out[: a.top]:
        print("%8.3f ms %4d x avg %7.1f us  %-18s <%s> wgs=%d lds=%d" % (tot * 1e-6, cnt, tot / cnt * 1e-3, base, targs, wgs, lds))


if __name__ == "__main__":
    main()
